"""Oracle pin for object instancing (core/src/primitives/transformed_primitive.rs, api/src/lib.rs:911-1000): a scene built from
ObjectInstances must see the same geometry as the same triangles transformed on the host and added directly ("flattened").  The two differ
only by floating-point rounding (and the t_max -= dt shift of transform_ray), so hits agree in primitive and instance identity and in t
to ~1e-3 relative, occlusion agrees, and the rendered images agree to Monte Carlo-free tolerance when nothing is stochastic."""
import numpy as np
import pytest

import scenes
from oracle_binding import OracleScene

I4 = (np.eye(4, dtype=np.float32).reshape(16),) * 2


def _build(host, instanced):
    P, idx = host.gen_random_tris(300, 5)
    Pg, ig = scenes.grid_mesh(4, z=-1.2, size=3.0)
    mul = host.compose
    T = [mul(mul(I4, host.translate([2.5, 0.3, -0.2])), host.rotate(40, [0.2, 1, 0.3])),
         mul(mul(I4, host.translate([-2.0, 0.5, 0.4])), host.scale([0.7, 1.3, 0.9])), I4]
    o = OracleScene()
    m = o.add_material_matte((0.5, 0.5, 0.5), 0.0)
    o.add_mesh(Pg, ig, m)
    if instanced:
        ob = o.object_begin(); o.add_mesh(P, idx, m); o.object_end()
        one = o.object_begin(); o.add_mesh(P[:3] * np.float32(2.0), [0, 1, 2], m); o.object_end()
        o.add_instance(ob, *T[0]); o.add_instance(ob, *T[1]); o.add_instance(one, *T[0]); o.add_instance(ob, *T[2])
    else:
        o.add_mesh(host.transform_points(T[0][0], P), idx, m); o.add_mesh(host.transform_points(T[1][0], P), idx, m)
        o.add_mesh(host.transform_points(T[0][0], P[:3] * np.float32(2.0)), [0, 1, 2], m); o.add_mesh(P, idx, m)
    o.build_accel(0, 4)
    return o, len(ig) // 3


def test_instanced_scene_matches_flattened_scene(host):
    a, ntg = _build(host, True)
    b, _ = _build(host, False)
    rays = scenes.random_rays(20000, 3, bound=3.0)
    ha, _ = a.intersect_batch_stats(rays); hb, _ = b.intersect_batch_stats(rays)
    hit_a = ha["prim"] != 0xFFFFFFFF; hit_b = hb["prim"] != 0xFFFFFFFF
    assert (hit_a != hit_b).mean() < 2e-4            # only grazing rays may flip
    both = hit_a & hit_b
    assert np.max(np.abs(ha["t"][both] - hb["t"][both]) / np.maximum(hb["t"][both], 1e-3)) < 2e-3
    k = hb["prim"][both].astype(np.int64) - ntg       # flattened prim -> (instance number + 1, triangle inside the object)
    inst_b = np.where(k < 0, 0, np.where(k < 300, 1, np.where(k < 600, 2, np.where(k < 601, 3, 4))))
    tri_b = np.where(k < 0, hb["prim"][both], np.where(k < 300, k, np.where(k < 600, k - 300, np.where(k < 601, 0, k - 601))))
    same = (ha["pad"][both, 1] == inst_b)
    assert same.mean() > 0.9995
    tri_a = np.where(ha["pad"][both, 1] == 0, ha["prim"][both], np.where(ha["pad"][both, 1] == 3, 0, ha["prim"][both].astype(np.int64) - ntg))
    assert (tri_a[same] == tri_b[same]).mean() > 0.9995
    oa, _ = a.occluded_batch_stats(rays); ob_, _ = b.occluded_batch_stats(rays)
    assert (oa != ob_).mean() < 2e-4
    # TransformedPrimitive::world_bound = transform_bounds of the object's box: never tighter than the flattened triangles' box
    wa, wb = a.world_bound(), b.world_bound()
    assert (wa[:3] <= wb[:3] + 1e-5).all() and (wa[3:] >= wb[3:] - 1e-5).all()


def test_area_light_inside_an_object_is_emission_without_a_light(host):
    """api/src/lib.rs:877-881 ("Area lights not supported with object instancing"): the shapes of an object definition keep their area light — a camera ray or a specular chain that
    reaches one sees its emission, SurfaceInteraction::le — but the light never joins Scene::lights, so nothing samples it and no diffuse surface is lit by it.
    Pin: against the same scene without the AreaLightSource, pixels differ only where the emitter is seen directly or in the mirror, there by its radiance L; the matte floor
    next to it receives nothing from it."""
    from emissive_object_scene import emissive_object_scene, L_EMIT
    a = OracleScene(); emissive_object_scene(a, host, emissive=True)
    b = OracleScene(); emissive_object_scene(b, host, emissive=False)
    xa, wa, sa, _ = a.render_path_ex(max_depth=4, threads=4); xb, wb, sb, _ = b.render_path_ex(max_depth=4, threads=4)
    assert (sa.regular_rays, sa.shadow_rays) == (sb.regular_rays, sb.shadow_rays)      # same paths: no light was added, no sampling decision changed
    ra, rb = a.film_to_rgb(xa, wa), b.film_to_rgb(xb, wb)
    d = ra - rb
    lit = np.abs(d).max(axis=2) > 1e-6
    assert 20 < lit.sum() < 0.4 * lit.size                                             # the two quads and their mirror images, nothing else
    full = np.abs(d - np.float32(L_EMIT)).max(axis=2) < 1e-3                          # pixels whose every sample looks at the emitter: exactly its radiance on top of the black surface
    assert full.sum() >= 10 and (full <= lit).all()
    assert (d[lit] >= -1e-6).all() and (d[lit] <= np.float32(L_EMIT) * 1.001 + 1e-5).all()   # partial coverage and the 0.9 mirror: a fraction of L, never more
    # the floor right under the emitters is lit by the distant light alone in both renders: no radiance arrives from the dropped light
    floor_rows = slice(int(0.8 * ra.shape[0]), ra.shape[0])
    assert np.array_equal(ra[floor_rows], rb[floor_rows]) and float(ra[floor_rows].mean()) > 0.01
