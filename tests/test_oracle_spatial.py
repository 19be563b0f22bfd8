"""Oracle pinning for SpatialLightDistribution (core/src/light_distrib/spatial.rs): the restatement is checked against
closed forms evaluated independently in float64 (voxel resolution rule, 128-point Halton quadrature of delta lights, the
min_contrib floor, Distribution1D normalisation).  No reference test or fixture exists for this file."""
import numpy as np
import pytest

from oracle_binding import OracleScene


def radical_inverse(base, i):
    inv, f, r = 1.0 / base, 1.0 / base, 0.0
    while i:
        r += (i % base) * f
        i //= base
        f *= inv
    return r


def box_scene(o, host, size=(4.0, 2.0, 1.0), lights=()):
    """An axis-aligned slab of 12 triangles spanning [0,size]; lights added first (ids 0..)."""
    for kind, args in lights:
        getattr(o, "add_light_" + kind)(*args)
    sx, sy, sz = size
    P = np.array([[x, y, z] for x in (0, sx) for y in (0, sy) for z in (0, sz)], np.float32)
    idx = np.array([0, 1, 3, 0, 3, 2, 4, 6, 7, 4, 7, 5, 0, 4, 5, 0, 5, 1, 2, 3, 7, 2, 7, 6, 0, 2, 6, 0, 6, 4, 1, 5, 7, 1, 7, 3], np.uint32)
    mat = o.add_material_matte((0.5, 0.5, 0.5), 0.0)
    o.add_mesh(P, idx, mat)
    w2c, c2w = host.look_at([2, -6, 0.5], [2, 1, 0.5], [0, 0, 1])
    o.set_camera_perspective(host.perspective_raster_to_camera(50.0, 16, 16), c2w)
    cb, table, sb = host.film_box(16, 16)
    o.set_film(16, 16, cb, (0.5, 0.5), table)
    o.set_sampler(0, 2, sb)
    o.build_accel(0, 4)
    return sb


def test_voxel_resolution_rule(host):
    # widest axis gets 64 voxels, the others round(diag/bmax*64), at least 1 (spatial.rs:63-75)
    with OracleScene() as o:
        box_scene(o, host, size=(4.0, 2.0, 0.01), lights=[("point", ((1, 1, 1), (2, 1, 3))), ("point", ((1, 1, 1), (0, 0, 3)))])
        o.render_path(max_depth=1, light_strategy=2)
        nv, created = o.spatial_stats()
    assert nv == (64, 32, 1)
    assert created > 0


def test_delta_lights_quadrature_and_floor(host):
    I = np.float32([3.0, 2.0, 1.0]); Ld = np.float32([0.5, 0.25, 1.0])
    pl = np.array([1.0, 1.0, 3.0])
    y = lambda c: 0.212671 * c[0] + 0.715160 * c[1] + 0.072169 * c[2]
    with OracleScene() as o:
        # light 2: a one-sided emissive triangle far below the slab, facing down: L(n, w) = 0 for every point of the slab
        lights = [("point", (I, pl.astype(np.float32))), ("distant", (Ld, np.float32([0, 0, 1])))]
        box_scene(o, host, size=(4.0, 2.0, 1.0), lights=lights)
        func, cdf, func_int = o.spatial_voxel((10, 5, 3), 2)
    nv = (64, 32, 16)
    lo = np.array([10 / 64 * 4.0, 5 / 32 * 2.0, 3 / 16 * 1.0]); hi = np.array([11 / 64 * 4.0, 6 / 32 * 2.0, 4 / 16 * 1.0])
    acc = 0.0
    for i in range(128):
        t = np.array([radical_inverse(2, i), radical_inverse(3, i), radical_inverse(5, i)])
        p = (1 - t) * lo + t * hi
        acc += y(I) / np.sum((pl - p) ** 2)
    assert func[0] == pytest.approx(acc, rel=2e-5)
    assert func[1] == pytest.approx(128 * y(Ld), rel=1e-6)
    # Distribution1D::new: cdf[i] = sum func/n, normalised; func_int = mean(func)
    assert func_int == pytest.approx((func[0] + func[1]) / 2, rel=1e-6)
    assert cdf[0] == 0 and cdf[2] == 1 and cdf[1] == pytest.approx(func[0] / (func[0] + func[1]), rel=1e-6)


def test_min_contrib_floor(host):
    """A light that contributes nothing in a voxel still gets 0.001 * average contribution (spatial.rs:141-150)."""
    L = np.float32([4, 4, 4])
    with OracleScene() as o:
        o.add_light_distant(np.float32([1, 1, 1]), np.float32([0, 0, 1]))
        lid = o.add_light_diffuse_area(L, 1, two_sided=False)
        mat = o.add_material_matte((0.5, 0.5, 0.5), 0.0)
        # emissive triangle below the slab whose normal points down (-z): nothing above it is lit
        o.add_mesh(np.array([[0, 0, -2], [0, 1, -2], [1, 0, -2]], np.float32), [0, 1, 2], mat, first_area_light=lid)
        box_scene(o, host, size=(4.0, 2.0, 1.0))
        nv_probe = (5, 5, 40)  # well above the emitter in the 64-voxel z axis of [-2, 1]
        func, cdf, func_int = o.spatial_voxel(nv_probe, 2)
    y1 = 0.212671 + 0.715160 + 0.072169
    assert func[0] == pytest.approx(128 * y1, rel=1e-6)
    avg = func[0] / (128 * 2)
    assert func[1] == pytest.approx(0.001 * avg, rel=1e-5)


def test_spatial_is_unbiased_against_uniform(host):
    """Same scene, uniform vs spatial light selection: the estimators agree in the mean (different variance)."""
    def capture(o):
        lights = [("point", (np.float32([8, 8, 8]), np.float32([1, 1, 3]))), ("point", (np.float32([1, 1, 1]), np.float32([3.5, 1.5, 2]))),
                  ("infinite", (np.float32([0.3, 0.3, 0.3]),))]
        return box_scene(o, host, size=(4.0, 2.0, 1.0), lights=lights)
    means = []
    for strat in (0, 2):
        with OracleScene() as o:
            capture(o)
            o.set_sampler(0, 64, [0, 0, 16, 16])
            xyz, wt, st = o.render_path(max_depth=3, light_strategy=strat)
            means.append(o.film_to_rgb(xyz, wt).mean())
            if strat == 2:
                assert st.light_distributions_created == o.spatial_stats()[1] > 0
    assert means[1] == pytest.approx(means[0], rel=0.03)
