"""-m gpu: mirror / plastic / glass (smooth and rough) / metal / uber on the device (csrc/bsdf_general.h, shade_kernel<true>) against the
oracle's restatement of core/src/reflection/* and materials/src/*.rs: film bit-exact in f64-libm mode, same ray and path counters."""
import numpy as np
import pytest

import pbrt_hip
import scenes
from oracle_binding import OracleScene, set_libm_mode

pytestmark = pytest.mark.gpu


def quad(a, b, c, d):
    return np.array([a, b, c, d], np.float32), np.array([0, 1, 2, 0, 2, 3], np.uint32)


def box_mesh(lo, hi):
    """12 outward-facing triangles of an axis-aligned box (a closed dielectric needs consistent normals)."""
    x0, y0, z0 = lo; x1, y1, z1 = hi
    P = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]], np.float32)
    idx = np.array([0, 2, 1, 0, 3, 2, 4, 5, 6, 4, 6, 7, 0, 1, 5, 0, 5, 4, 2, 3, 7, 2, 7, 6, 1, 2, 6, 1, 6, 5, 3, 0, 4, 3, 4, 7], np.uint32)
    return P, idx


def material_scene(host, make_materials, res=40, spp=8, sigma=0.0):
    """A room lit by an area light, a point light and an environment; `make_materials(s)` returns (floor, wall, slab, panel, block)."""
    def cap(s):
        s.add_light_infinite((0.3, 0.35, 0.4))
        s.add_light_point((4, 4, 5), (-0.7, -0.9, 0.8))
        floor, wall, slab, panel, block = make_materials(s)
        white = s.add_material_matte((0.7, 0.7, 0.7), sigma)
        for k, (P, idx) in enumerate([quad([-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1]), quad([-1, 1, -1], [1, 1, -1], [1, 1, 1], [-1, 1, 1]),
                                      quad([-1, -1, -1], [-1, 1, -1], [-1, 1, 1], [-1, -1, 1]), quad([1, -1, -1], [1, -1, 1], [1, 1, 1], [1, 1, -1]),
                                      quad([-1, -1, 1], [-1, 1, 1], [1, 1, 1], [1, -1, 1])]):
            s.add_mesh(P, idx, [floor, wall, white, wall, white][k])
        lid = s.add_light_diffuse_area((9.0, 8.0, 7.0), 2, two_sided=False)
        P, idx = quad([-0.3, -0.3, 0.98], [0.3, -0.3, 0.98], [0.3, 0.3, 0.98], [-0.3, 0.3, 0.98])
        s.add_mesh(P, idx, white, first_area_light=lid, reverse_orientation=True)
        P, idx = box_mesh((-0.6, -0.3, -0.7), (-0.1, 0.1, 0.0)); s.add_mesh(P, idx, slab)      # closed solid (refraction in and out)
        P, idx = quad([0.2, 0.4, -0.8], [0.8, 0.1, -0.8], [0.8, 0.3, 0.2], [0.2, 0.6, 0.2]); s.add_mesh(P, idx, panel)
        P, idx = box_mesh((0.1, -0.6, -1.0), (0.6, -0.2, -0.5)); s.add_mesh(P, idx, block)
        w2c, c2w = host.look_at([0.1, -3.3, 0.1], [0, 0, -0.1], [0, 0, 1])
        s.set_camera_perspective(host.perspective_raster_to_camera(42.0, res, res), c2w)
        cb, table, sb = host.film_box(res, res)
        s.set_film(res, res, cb, (0.5, 0.5), table)
        s.set_sampler(0, spp, sb)
        s.build_accel(0, 4)
    return cap


def _check(cap, strategy=0, max_depth=6):
    prod = pbrt_hip.Scene(); orc = OracleScene()
    cap(prod); cap(orc)
    set_libm_mode(1)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(max_depth=max_depth, light_strategy=strategy)
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(max_depth=max_depth, light_strategy=strategy)
    assert (gst.regular_rays, gst.shadow_rays, gst.paths_total, gst.paths_zero_radiance) == \
           (ost.regular_rays, ost.shadow_rays, ost.paths_total, ost.paths_zero_radiance), (gst.as_dict(), ost.as_dict())
    assert np.array_equal(gwt.view(np.uint32), owt.view(np.uint32))
    diff = (gxyz.view(np.uint32) != oxyz.view(np.uint32)).any(axis=2)
    assert int(diff.sum()) == 0, f"{int(diff.sum())} pixels differ, max abs {np.abs(gxyz - oxyz).max()}"
    assert np.isfinite(oxyz).all() and float(oxyz.max()) > 0
    return oxyz


MATERIAL_SETS = {
    "mirror_glass": lambda s: (s.add_material_matte((0.5, 0.5, 0.5), 0.0), s.add_material_mirror((0.9, 0.85, 0.8)), s.add_material_glass((1, 1, 1), (1, 1, 1), 0, 0, 1.5),
                               s.add_material_mirror((0.7, 0.7, 0.9)), s.add_material_glass((0.9, 1, 1), (1, 0.9, 0.8), 0, 0, 1.33)),
    "plastic_metal": lambda s: (s.add_material_plastic((0.3, 0.25, 0.2), (0.3, 0.3, 0.3), 0.15, True), s.add_material_matte((0.6, 0.2, 0.2), 25.0),
                                s.add_material_metal((0.2, 0.92, 1.1), (3.9, 2.45, 2.14), 0.08, 0.08, True), s.add_material_plastic((0.1, 0.4, 0.6), (0.5, 0.5, 0.5), 0.02, False),
                                s.add_material_metal((1.65, 0.88, 0.52), (9.2, 6.3, 4.8), 0.2, 0.05, True)),
    "rough_glass_uber": lambda s: (s.add_material_uber((0.4, 0.4, 0.3), (0.2, 0.2, 0.2), (0.1, 0.1, 0.1), (0, 0, 0), (1, 1, 1), 0.1, 0.1, 1.5, True),
                                   s.add_material_matte((0.5, 0.6, 0.5), 0.0), s.add_material_glass((1, 1, 1), (1, 1, 1), 0.1, 0.25, 1.5, True),
                                   s.add_material_uber((0.3, 0.1, 0.1), (0.3, 0.3, 0.3), (0.2, 0.2, 0.2), (0.4, 0.4, 0.4), (0.7, 0.6, 0.5), 0.05, 0.2, 1.3, True),
                                   s.add_material_glass((1, 1, 1), (0.8, 0.9, 1.0), 0.3, 0.3, 1.6, False)),
    "substrate_translucent_mix": lambda s: (s.add_material_substrate((0.5, 0.4, 0.3), (0.3, 0.3, 0.3), 0.15, 0.05, True),
                                            s.add_material_mix(s.add_material_matte((0.7, 0.2, 0.2), 0.0), s.add_material_mirror((0.9, 0.9, 0.9)), (0.7, 0.6, 0.5)),
                                            s.add_material_translucent((0.4, 0.5, 0.4), (0.3, 0.3, 0.3), (0.5, 0.5, 0.5), (0.6, 0.5, 0.4), 0.1, True),
                                            s.add_material_mix(s.add_material_mix(s.add_material_plastic(), s.add_material_glass(), (0.5, 0.5, 0.5)),
                                                               s.add_material_substrate(), (0.3, 0.6, 0.9)),
                                            s.add_material_translucent((0.25,) * 3, (0, 0, 0), (0, 0, 0), (0.9, 0.9, 0.9), 0.1, False)),
}


@pytest.mark.parametrize("name", list(MATERIAL_SETS))
def test_material_sets_film_bit_exact(host, name):
    _check(material_scene(host, MATERIAL_SETS[name]))


def test_materials_with_spatial_strategy_and_deep_paths(host):
    _check(material_scene(host, MATERIAL_SETS["mirror_glass"], res=32, spp=4), strategy=2, max_depth=12)


def test_matte_only_scene_is_identical_under_both_kernels(host):
    """A scene of matte materials rendered by shade_kernel<false>; adding an unused plastic material switches to shade_kernel<true>:
    the general BSDF must reproduce the matte fast path bit for bit."""
    base = scenes.cornell_like(host, with_normals=True, sigma=20.0)

    def with_unused(s):
        s.add_material_plastic()
        base(s)
    with pbrt_hip.Scene() as a, pbrt_hip.Scene() as b:
        base(a); with_unused(b)
        xa, wa, sa = a.render_path(max_depth=5, light_strategy=1)
        xb, wb, sb = b.render_path(max_depth=5, light_strategy=1)
    assert np.array_equal(xa.view(np.uint32), xb.view(np.uint32)) and (sa.regular_rays, sa.shadow_rays) == (sb.regular_rays, sb.shadow_rays)


@pytest.mark.parametrize("name", list(MATERIAL_SETS))
def test_material_sets_within_tolerance_of_glibc_libm(host, name):
    """Against the oracle in libm mode 0 (glibc's f32 sin/cos/acos/atan2 = what the Rust reference links) the film is no longer bit-equal: a
    last-bit difference in a trig call can flip a discrete decision (lobe choice, Russian roulette) on rare paths.  Stated tolerance for scenes
    with specular / glossy materials at 64^2 x 32 spp: RMSE <= 2e-3 x mean luminance, <= 0.2 % of pixels off by more than 1e-2 x mean."""
    cap = material_scene(host, MATERIAL_SETS[name], res=64, spp=32)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    cap(prod); cap(orc)
    set_libm_mode(0)
    oxyz, owt, ost, _ = orc.render_path_ex(max_depth=6, light_strategy=0)
    gxyz, gwt, gst = prod.render_path(max_depth=6, light_strategy=0)
    g = prod.film_to_rgb(gxyz, gwt); o = prod.film_to_rgb(oxyz, owt)
    mean = float(o.mean()); d = g - o
    assert np.sqrt((d ** 2).mean()) / mean <= 2e-3
    assert float((np.abs(d).max(axis=2) > 1e-2 * mean).mean()) <= 2e-3
    assert abs(int(gst.regular_rays) - int(ost.regular_rays)) <= 1e-4 * ost.regular_rays


def test_none_material_veils_bit_exact(host):
    """Material "none" (null BSDF, path.rs:142-150) in front of and inside the material scene: paths need more wavefront rounds than
    max_depth + 1; film and counters still equal the oracle's, with the spatial strategy too (no voxel lookup at skipped surfaces)."""
    base = material_scene(host, MATERIAL_SETS["plastic_metal"], res=32, spp=4)

    def cap(s):
        none = s.add_material_none()
        for y in (-2.6, -2.3, -1.5):
            P, idx = quad([-0.9, y, -0.9], [0.9, y, -0.9], [0.9, y, 0.9], [-0.9, y, 0.9]); s.add_mesh(P, idx, none)
        P, idx = quad([-0.9, -0.9, -0.2], [0.9, -0.9, -0.2], [0.9, 0.9, -0.2], [-0.9, 0.9, -0.2]); s.add_mesh(P, idx, none)
        base(s)
    for strategy, depth in ((0, 3), (2, 5), (1, 1)):
        _check(cap, strategy=strategy, max_depth=depth)


@pytest.mark.parametrize("n_materials", [700, 4300])
def test_more_materials_than_work_queue_bins_or_triangle_flag_bits(host, n_materials):
    """The shade-side work queues (csrc/matsort.h) key a path by its hit's material: 1 024 bins (materials beyond 500 textured / 500 plain share the last key of their
    group), and the material id travels with the hit in 12 bits of the TriRec's flags (ids from 4 095 on are looked up through the mesh).  A scene with 700 and one with 4 300
    materials — every triangle its own, every third one image-textured, plastic / matte alternating — renders the oracle's film bit for bit: sharing a key or taking the slow
    look-up changes who shades a path when, not what is computed."""
    from texture_scenes import make_image
    P, idx = host.gen_random_tris(n_materials, 31)
    img = make_image(32, 32, seed=3)

    def cap(s):
        s.add_light_infinite((0.6, 0.65, 0.7))
        s.add_light_point((6, 5, 5), (0.4, -2.5, 1.5))
        tex = s.add_texture_imagemap(s.add_mipmap(img), su=3.0, sv=2.0)
        r = np.random.default_rng(7)
        for t in range(n_materials):
            c = tuple(float(v) for v in r.uniform(0.2, 0.9, 3))
            if t % 3 == 0:
                m = s.add_material_matte_tex(tex, float(r.uniform(0, 30)))
            elif t % 2:
                m = s.add_material_plastic(c, (0.2, 0.2, 0.2), 0.1, True)
            else:
                m = s.add_material_matte(c, 0.0)
            s.add_mesh(P[3 * t:3 * t + 3], np.array([0, 1, 2], np.uint32), m)
        w2c, c2w = host.look_at([0, -4, 0], [0, 0, 0], [0, 0, 1])
        s.set_camera_perspective(host.perspective_raster_to_camera(40.0, 48, 48), c2w)
        cb, table, sb = host.film_box(48, 48)
        s.set_film(48, 48, cb, (0.5, 0.5), table)
        s.set_sampler(0, 8, sb)
        s.build_accel(0, 4)
    _check(cap, strategy=0, max_depth=4)
