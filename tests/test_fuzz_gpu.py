"""-m gpu: randomized differential test.  Each seed draws a scene from everything the library supports at once — split method,
instanced objects, all material kinds with random parameters, all light kinds, per-vertex N / UV, alpha masks, thin-lens camera,
any pixel filter, Halton sampler, every light-sampling strategy, odd resolutions / crops — and the film must equal the oracle's bit for
bit (f64-libm mode) together with the ray and path counters.  Failures print the seed."""
import numpy as np
import pytest

import pbrt_hip
import scenes
from oracle_binding import OracleScene, set_libm_mode

pytestmark = pytest.mark.gpu
I4 = (np.eye(4, dtype=np.float32).reshape(16),) * 2


def random_material(s, rng):
    k = rng.integers(0, 11)
    c = lambda lo=0.05, hi=0.95: tuple(rng.uniform(lo, hi, 3).astype(np.float32))
    if k == 0:
        return s.add_material_matte(c(), float(rng.choice([0.0, rng.uniform(1, 60)])))
    if k == 1:
        return s.add_material_mirror(c(0.3, 1.0))
    if k == 2:
        return s.add_material_plastic(c(), c(0.05, 0.5), float(rng.uniform(0.01, 0.4)), bool(rng.integers(0, 2)))
    if k == 3:
        rough = rng.integers(0, 2)
        return s.add_material_glass(c(0.5, 1), c(0.5, 1), float(rough * rng.uniform(0.02, 0.3)), float(rough * rng.uniform(0.02, 0.3)), float(rng.uniform(1.1, 1.8)), True)
    if k == 4:
        return s.add_material_metal(c(0.1, 2.0), c(1.5, 6.0), float(rng.uniform(0.01, 0.3)), float(rng.uniform(0.01, 0.3)), bool(rng.integers(0, 2)))
    if k == 5:
        op = float(rng.choice([1.0, rng.uniform(0.3, 0.9)]))
        return s.add_material_uber(c(), c(0.05, 0.4), c(0, 0.3), c(0, 0.3), (op, op, op), float(rng.uniform(0.02, 0.3)), float(rng.uniform(0.02, 0.3)), float(rng.uniform(1.1, 1.7)), True)
    if k == 6:
        return s.add_material_substrate(c(), c(0.05, 0.6), float(rng.uniform(0.02, 0.4)), float(rng.uniform(0.02, 0.4)), bool(rng.integers(0, 2)))
    if k == 7:
        return s.add_material_translucent(c(), c(0, 0.5), c(0.1, 0.9), c(0, 0.9), float(rng.uniform(0.02, 0.3)), True)
    if k == 8:
        a = s.add_material_plastic(c(), c(0.05, 0.5), float(rng.uniform(0.01, 0.4)), True)
        b = s.add_material_mirror(c(0.3, 1.0)) if rng.integers(0, 2) else s.add_material_glass(c(0.5, 1), c(0.5, 1), 0.0, 0.0, 1.5, True)
        return s.add_material_mix(a, b, c(0.1, 0.9))
    if k == 9:
        return s.add_material_none()              # null BSDF: surfaces are passed through (path.rs:142-150)
    return s.add_material_matte((0, 0, 0), 0.0)   # black: no BxDF at all (matte.rs:66)


def random_transform(host, rng, spread=1.5):
    t = host.compose(I4, host.translate(rng.uniform(-spread, spread, 3)))
    t = host.compose(t, host.rotate(float(rng.uniform(0, 360)), rng.normal(size=3) + 1e-3))
    sc = rng.uniform(0.4, 1.3, 3) * rng.choice([1.0, 1.0, -1.0], 3)
    return host.compose(t, host.scale(sc))


def build_case(host, seed, big=False):
    """big: a few hundred thousand paths over thousands of triangles (many shade blocks, several traversal batches per wave)."""
    rng = np.random.default_rng(seed)
    res = (int(rng.integers(17, 41)), int(rng.integers(13, 37)))
    spp = int(rng.choice([1, 2, 3, 5, 8]))
    if big:
        res = (int(rng.integers(120, 200)), int(rng.integers(90, 160))); spp = int(rng.choice([4, 8, 12]))
    split = int(rng.choice([0, 0, 0, 3]))
    fkind = str(rng.choice(["box", "gaussian", "mitchell", "triangle", "sinc"]))
    radius = {"box": (0.5, 0.5), "gaussian": (1.5, 2.0), "mitchell": (2.0, 2.0), "triangle": (1.0, 2.0), "sinc": (3.0, 2.5)}[fkind]
    fparams = {"box": (0, 0), "gaussian": (2.0, 0), "mitchell": (1 / 3, 1 / 3), "triangle": (0, 0), "sinc": (3.0, 0)}[fkind]
    crop = (0.0, 1.0, 0.0, 1.0) if rng.integers(0, 2) else tuple(np.sort(rng.uniform(0, 1, 2))) + tuple(np.sort(rng.uniform(0, 1, 2)))
    n_lights = int(rng.integers(0, 4))
    light_rng_state = rng.integers(0, 2 ** 31)
    geo_seed = int(rng.integers(0, 2 ** 31))
    lens = float(rng.choice([0.0, 0.05]))
    use_inst = bool(rng.integers(0, 2))

    def cap(s):
        lr = np.random.default_rng(light_rng_state)
        g = np.random.default_rng(geo_seed)
        if lr.integers(0, 2):
            t = random_transform(host, lr)
            s.add_light_infinite(tuple(lr.uniform(0.1, 0.8, 3)), t[0], t[1])
        for _ in range(n_lights):
            k = lr.integers(0, 3)
            if k == 0:
                s.add_light_point(tuple(lr.uniform(2, 12, 3)), lr.uniform(-1.5, 1.5, 3).astype(np.float32))
            elif k == 2:
                t = random_transform(host, lr, 0.5)
                l2w, w2l, ctw, cfs = host.spot(t, lr.uniform(-1, 1, 3) + np.array([0, 0, 1.5]), lr.uniform(-0.5, 0.5, 3), float(lr.uniform(15, 70)), float(lr.uniform(1, 14)))
                s.add_light_spot(tuple(lr.uniform(5, 30, 3)), l2w, w2l, ctw, cfs)
            else:
                w = lr.normal(size=3); w /= np.linalg.norm(w)
                s.add_light_distant(tuple(lr.uniform(0.3, 2, 3)), np.float32(w))
        mats = [random_material(s, g) for _ in range(4)]
        # emissive mesh (area lights), scene-level
        P, idx = host.gen_random_tris(int(g.integers(1, 6)), int(g.integers(1, 1000)))
        lid = s.add_light_diffuse_area(tuple(g.uniform(2, 10, 3)), len(idx) // 3, two_sided=bool(g.integers(0, 2)))
        s.add_mesh(P * np.float32(0.4) + np.float32([0, 0, 1.3]), idx, mats[0], first_area_light=lid, reverse_orientation=bool(g.integers(0, 2)))
        # scene-level meshes with optional N / UV / alpha
        for k in range(int(g.integers(1, 4))):
            P, idx = host.gen_random_tris(int(g.integers(5, 120)) * (60 if big else 1), int(g.integers(1, 1000)))
            N = g.normal(size=P.shape).astype(np.float32) if g.integers(0, 2) else None
            UV = g.uniform(0, 1, (len(P), 2)).astype(np.float32) if g.integers(0, 2) else None
            s.add_mesh(P, idx, mats[k % 4], N=N, UV=UV, reverse_orientation=bool(g.integers(0, 2)), swaps_handedness=bool(g.integers(0, 2)),
                       alpha=float(g.choice([1.0, 1.0, 0.0])), shadow_alpha=float(g.choice([1.0, 1.0, 0.0])))
        Pg, ig = scenes.grid_mesh(3, z=-1.3, size=2.5)
        s.add_mesh(Pg, ig, mats[3])
        if use_inst:
            ob = s.object_begin()
            P, idx = host.gen_random_tris(int(g.integers(2, 60)), int(g.integers(1, 1000)))
            s.add_mesh(P * np.float32(0.5), idx, mats[1], N=(g.normal(size=P.shape).astype(np.float32) if g.integers(0, 2) else None))
            s.object_end()
            one = s.object_begin(); s.add_mesh(P[:3], [0, 1, 2], mats[2]); s.object_end()
            for _ in range(int(g.integers(1, 5))):
                t = random_transform(host, g)
                s.add_instance(ob if g.integers(0, 3) else one, t[0], t[1])
        w2c, c2w = host.look_at(g.uniform(-0.5, 0.5, 3) + np.array([0, -4.5, 0.5]), [0, 0, 0], [0, 0, 1])
        s.set_camera_perspective(host.perspective_raster_to_camera(float(g.uniform(30, 60)), res[0], res[1]), c2w, lens_radius=lens, focal_distance=4.5)
        cb, table, sb = host.film_filter(fkind, res[0], res[1], radius, fparams, crop)
        s.set_film(res[0], res[1], cb, radius, table, scale=1.0, max_sample_luminance=float(g.choice([np.inf, 5.0])))
        s.set_sampler(0, spp, sb)
        s.build_accel_best(split, int(g.choice([1, 4, 8])))   # the product builds on the GPU where it can (same tree); the oracle has one builder
        return cb
    return cap, dict(max_depth=int(rng.integers(1, 9)), light_strategy=int(rng.integers(0, 3)), rr_threshold=float(rng.choice([1.0, 0.5, 10.0])))


@pytest.mark.parametrize("seed", list(range(1, 25)))
def test_random_scene_film_bit_exact(host, seed):
    cap, kw = build_case(host, seed)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    cb = cap(prod); cap(orc)
    if (cb[2] - cb[0]) * (cb[3] - cb[1]) <= 0:
        pytest.skip("empty crop window")
    set_libm_mode(1)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(**kw)
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(**kw)
    assert (gst.regular_rays, gst.shadow_rays, gst.paths_total, gst.paths_zero_radiance, gst.light_distributions_created) == \
           (ost.regular_rays, ost.shadow_rays, ost.paths_total, ost.paths_zero_radiance, ost.light_distributions_created), (seed, kw, gst.as_dict(), ost.as_dict())
    assert np.array_equal(gwt.view(np.uint32), owt.view(np.uint32)), seed
    nb = int((gxyz.view(np.uint32) != oxyz.view(np.uint32)).any(axis=2).sum())
    assert nb == 0, f"seed {seed} {kw}: {nb} pixels differ, max abs {np.abs(gxyz - oxyz).max()}"


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_random_scene_medium_size(host, seed):
    """The same generator at a few hundred thousand paths: many shade blocks per launch, sample chunks, block-local sorting across blocks."""
    cap, kw = build_case(host, seed, big=True)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    cb = cap(prod); cap(orc)
    if (cb[2] - cb[0]) * (cb[3] - cb[1]) <= 0:
        pytest.skip("empty crop window")
    set_libm_mode(1)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(**kw)
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(**kw)
    assert (gst.regular_rays, gst.shadow_rays, gst.paths_total, gst.paths_zero_radiance, gst.light_distributions_created) == \
           (ost.regular_rays, ost.shadow_rays, ost.paths_total, ost.paths_zero_radiance, ost.light_distributions_created), (seed, kw)
    assert np.array_equal(gwt.view(np.uint32), owt.view(np.uint32)) and np.array_equal(gxyz.view(np.uint32), oxyz.view(np.uint32)), seed
