"""Pins of the CPU oracle (not gpu): known-answer vectors and analytic cases.

What the reference offers for this path: its #[test]s cover only core/src/geometry — replayed in
test_oracle_geometry.py — plus the constants of SURVEY Appendix C (tests/golden/kat_appendix_c.json).  Everything else
here is an analytic known answer the oracle must satisfy."""
import ctypes as C
import json
import math
import os
import re

import numpy as np
import pytest

import pbrt_hip
from oracle_binding import OracleScene, oracle_binding

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "kat_appendix_c.json")) as f:
        return json.load(f)


def _rng(lib, seq, default, n=4):
    out = np.zeros(n, np.uint32)
    lib.oracle_rng_u32(seq, 1 if default else 0, out.ctypes.data, n)
    return [f"0x{v:08x}" for v in out]


def test_pcg32_known_answers(kat):
    lib = oracle_binding().lib
    assert _rng(lib, 0, True) == kat["rng_default_u32"]
    for seq, want in kat["rng_seq"].items():
        assert _rng(lib, int(seq), False) == want


def test_halton_permutation_known_answers(kat):
    lib = oracle_binding().lib
    primes = [lib.oracle_prime(i) for i in range(6)]
    assert primes == [2, 3, 5, 7, 11, 13]
    for i, p in enumerate(primes):
        out = np.zeros(p, np.uint16)
        lib.oracle_halton_perm(i, out.ctypes.data)
        assert out.tolist() == kat["halton_perms"][str(p)]
        assert sorted(out.tolist()) == list(range(p))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "core/src/low_discrepency.rs")), reason="reference tree not present")
def test_prime_tables_match_reference_text():
    """PRIMES / PRIME_SUMS (core/src/low_discrepency.rs:13,102) are data; the oracle regenerates them with a sieve."""
    src = open(os.path.join(REF, "core/src/low_discrepency.rs")).read()
    lib = oracle_binding().lib
    for name, fn in (("PRIMES", lib.oracle_prime), ("PRIME_SUMS", lib.oracle_prime_sum)):
        m = re.search(r"pub const %s: \[usize; PRIME_TABLE_SIZE\] = \[(.*?)\];" % name, src, re.S)
        assert m, name
        vals = [int(v) for v in re.findall(r"\d+", m.group(1))]
        assert len(vals) == 1000
        assert vals == [fn(i) for i in range(1000)]


def test_radical_inverse_rationals():
    lib = oracle_binding().lib
    f32 = np.float32
    assert lib.oracle_radical_inverse(0, 1) == 0.5 and lib.oracle_radical_inverse(0, 2) == 0.25 and lib.oracle_radical_inverse(0, 3) == 0.75
    assert lib.oracle_radical_inverse(0, 0) == 0.0
    # base 3: digits reversed * 3^-k evaluated as the reference does (inv_base_n accumulated by repeated f32 multiply)
    inv3 = f32(1.0) / f32(3.0)
    assert f32(lib.oracle_radical_inverse(1, 1)) == f32(1) * inv3
    assert f32(lib.oracle_radical_inverse(1, 5)) == f32(7) * (inv3 * inv3)      # 5 = 12_3 -> 21_3 = 7
    inv5 = f32(1.0) / f32(5.0)
    assert f32(lib.oracle_radical_inverse(2, 7)) == f32(11) * (inv5 * inv5)     # 7 = 12_5 -> 21_5 = 11
    # every value in [0, 1)
    vals = [lib.oracle_radical_inverse(b, a) for b in range(0, 40) for a in (1, 2, 17, 12345, 2 ** 31 + 5)]
    assert all(0.0 <= v < 1.0 for v in vals)
    # scrambled: with index 0 only the infinite tail of perm[0] digits contributes: inv_base*perm[0]/(1-inv_base)
    for bi, p in ((2, 5), (3, 7), (4, 11)):
        perm = np.zeros(p, np.uint16); lib.oracle_halton_perm(bi, perm.ctypes.data)
        ib = f32(1.0) / f32(p)
        want = f32(1.0) * (f32(0.0) + ib * f32(perm[0]) / (f32(1.0) - ib))
        assert f32(lib.oracle_scrambled_radical_inverse(bi, 0)) == min(want, f32(0.99999994))


def test_halton_sampler_is_a_stratified_cover(host):
    """dims 0/1 of the 128x243-strided Halton index stratify each pixel: sample s of pixel (x,y) has film offset in [0,1)
    and the 64 samples of a pixel are distinct (samplers/src/halton.rs:118-160)."""
    s = OracleScene()
    pbrt_hip.capture_spec(pbrt_hip.SceneSpec(n_tris=4, xres=300, yres=200, spp=16), s, host)
    lib = oracle_binding().lib
    for (x, y) in ((0, 0), (5, 7), (127, 128), (299, 199)):
        pts = {(lib.oracle_sampler_value(s.h, x, y, k, 0), lib.oracle_sampler_value(s.h, x, y, k, 1)) for k in range(16)}
        assert len(pts) == 16 and all(0 <= u < 1 and 0 <= v < 1 for u, v in pts)
    # pixel (0,0), sample 0 is Halton index 0 -> film offset exactly (0,0) (the two-pixel add_sample case, film_tile.rs:70-74)
    assert lib.oracle_sampler_value(s.h, 0, 0, 0, 0) == 0.0 and lib.oracle_sampler_value(s.h, 128, 128, 0, 1) == 0.0


def _one_tri_scene(P):
    s = OracleScene()
    m = s.add_material_matte()
    s.add_mesh(P, [0, 1, 2], m)
    s.build_accel(0, 4)
    return s


def test_single_triangle_analytic_hit():
    P = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    s = _one_tri_scene(P)
    rays = np.zeros(4, pbrt_hip.RAY_DTYPE)
    rays["o"] = [[0.25, 0.25, 2.0], [0.25, 0.25, 2.0], [2.0, 2.0, 2.0], [0.25, 0.25, -1.0]]
    rays["d"] = [[0, 0, -1], [0, 0, -1], [0, 0, -1], [0, 0, 1]]
    rays["t_max"] = [np.inf, 1.5, np.inf, np.inf]
    h = s.intersect_batch(rays)
    assert h["prim"].tolist() == [0, 0xFFFFFFFF, 0xFFFFFFFF, 0]
    assert h["t"][0] == 2.0 and h["t"][3] == 1.0
    # p = b0*p0 + b1*p1 + b2*p2 = (0.25, 0.25, 0): b1 = b2 = 0.25, b0 = 0.5
    assert (h["b0"][0], h["b1"][0], h["b2"][0]) == (0.5, 0.25, 0.25)
    assert h["t"][1] == np.float32(1.5)  # miss keeps the ray's own t_max
    occ = s.occluded_batch(rays)
    assert occ.tolist() == [1, 0, 0, 1]


def test_box_slab_cases():
    lib = oracle_binding().lib

    def box(pmin, pmax, o, d, tmax=np.inf):
        a = np.array(list(pmin) + list(pmax) + list(o) + list(d) + [tmax], np.float32)
        out = np.zeros(4, np.float32)
        lib.oracle_geom_op(13, a.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
        return bool(out[0])
    assert box((-1, -1, -1), (1, 1, 1), (0, 0, 5), (0, 0, -1))
    assert not box((-1, -1, -1), (1, 1, 1), (0, 0, 5), (0, 0, 1))          # behind
    assert not box((-1, -1, -1), (1, 1, 1), (0, 0, 5), (0, 0, -1), 3.9)    # t_min = 4 >= t_max
    assert box((-1, -1, -1), (1, 1, 1), (0, 0, 5), (0, 0, -1), 4.1)
    assert box((-1, -1, -1), (1, 1, 1), (0, 0, 0), (1, 0, 0))              # origin inside
    assert not box((-1, -1, -1), (1, 1, 1), (3, 0, 5), (0, 0, -1))         # parallel, outside the x slab (inf/NaN arithmetic)
    # exactly ON the x = 1 plane with d.x = 0: (1-1)*inf = NaN poisons t_max and `t_max > 0` is false -> miss.  That is the
    # reference's comparison semantics (bounds3.rs:292-325), reproduced, not 'fixed'.
    assert not box((-1, -1, -1), (1, 1, 1), (1, 0, 5), (0, 0, -1))
    assert box((-1, -1, -1), (1, 1, 1), (0.999, 0, 5), (0, 0, -1))


def test_degenerate_triangle_is_never_hit():
    s = _one_tri_scene(np.array([[0, 0, 0], [1, 1, 0], [2, 2, 0]], np.float32))
    rays = np.zeros(1, pbrt_hip.RAY_DTYPE); rays["o"] = [[1, 1, 1]]; rays["d"] = [[0, 0, -1]]; rays["t_max"] = np.inf
    assert s.intersect_batch(rays)["prim"][0] == 0xFFFFFFFF and s.occluded_batch(rays)[0] == 0


def test_white_furnace(host):
    """Kd = 1 inside a constant environment L = 1: energy is conserved, so every pixel converges to 1 regardless of the
    geometry.  With NEE + MIS a finite-sample estimate is noisy but unbiased: check mean and spread."""
    spec = pbrt_hip.SceneSpec(n_tris=300, seed=5, xres=24, yres=24, spp=64, max_depth=60, kd=(1.0, 1.0, 1.0))
    s = OracleScene()
    pbrt_hip.capture_spec(spec, s, host)
    xyz, wt, st, _ = s.render_path_ex(max_depth=60, rr_threshold=0.0)   # rr_threshold 0 disables Russian roulette
    rgb = s.film_to_rgb(xyz, wt)
    assert abs(float(rgb.mean()) - 1.0) < 0.01, rgb.mean()
    assert float(rgb.min()) > 0.6 and float(rgb.max()) < 1.4  # Monte-Carlo spread at 64 spp


def test_grey_furnace_single_bounce_bound(host):
    """Kd = 0.5: radiance is 1 where the camera ray escapes and < 1 on geometry; the frame mean sits strictly between."""
    spec = pbrt_hip.SceneSpec(n_tris=300, seed=5, xres=24, yres=24, spp=16, max_depth=5)
    s = OracleScene()
    pbrt_hip.capture_spec(spec, s, host)
    xyz, wt, st, _ = s.render_path_ex()
    rgb = s.film_to_rgb(xyz, wt)
    assert 0.3 < float(rgb.mean()) < 1.0 and float(rgb.max()) <= 1.0001
    assert st.camera_rays == 24 * 24 * 16 and st.regular_rays >= st.camera_rays


def test_film_weight_sums_and_two_pixel_sample(host):
    """Box filter r = 0.5: every sample lands in exactly one pixel, except a sample whose film offset is exactly 0, which
    also reaches the pixel to its left/top (film_tile.rs:70-74 with ceil/floor): pixel (127,127) gets spp + 1."""
    spec = pbrt_hip.SceneSpec(n_tris=4, xres=130, yres=130, spp=4)
    s = OracleScene()
    pbrt_hip.capture_spec(spec, s, host)
    xyz, wt, st, _ = s.render_path_ex()
    assert wt[0, 0] == 4.0 and wt[5, 9] == 4.0
    assert wt[127, 127] == 5.0 and wt[127, 128] == 5.0 and wt[128, 127] == 5.0 and wt[128, 128] == 4.0


def test_tile_partition_is_exact(host):
    """Halton ignores the tile seed (halton.rs:177): rendering tiles in 3 parts and summing equals one full render bit for
    bit (the box filter's overlap rows only ever add exact zeros or a single non-zero term)."""
    spec = pbrt_hip.SceneSpec(n_tris=500, seed=2, xres=50, yres=37, spp=4)
    s = OracleScene()
    pbrt_hip.capture_spec(spec, s, host)
    full, wfull, _, _ = s.render_path_ex()
    acc = np.zeros_like(full); wacc = np.zeros_like(wfull)
    for p in range(3):
        x, w, _, _ = s.render_path_ex(tile_part=p, tile_parts=3)
        acc += x; wacc += w
    assert np.array_equal(wacc, wfull)
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))


def _sobol_tables():
    z = np.load(os.path.join(HERE, "golden", "sobol_subset.npz"))
    return z["m32"], z["vdc"], z["vdc_inv"]


def test_sobol_sampler_properties(host):
    """SobolSampler (samplers/src/sobol.rs:35-93): per pixel, the first 16 samples' film offsets form a (0,2)-net — one sample in
    each cell of the 4x4, 16x1 and 1x16 partitions of the pixel; every dimension stays in [0,1)."""
    s = OracleScene()
    spec = pbrt_hip.SceneSpec(n_tris=4, xres=100, yres=60, spp=16)
    pbrt_hip.capture_spec(spec, s, host)
    s.set_sobol_tables(*_sobol_tables())
    s.set_sampler(1, 16, s.sample_bounds)
    rays, pf = s.generate_camera_rays([0, 0, 100, 60], 0)
    assert np.isfinite(rays["d"]).all()
    offs = []
    for k in range(16):
        _, pf = s.generate_camera_rays([37, 21, 38, 22], k)
        offs.append(pf[0] - np.array([37, 21], np.float32))
    offs = np.array(offs)
    assert (offs >= 0).all() and (offs < 1).all()
    for nx, ny in ((4, 4), (16, 1), (1, 16), (2, 8), (8, 2)):
        cells = {(int(o[0] * nx), int(o[1] * ny)) for o in offs}
        assert len(cells) == 16, (nx, ny)
    xyz, wt, st, _ = s.render_path_ex(max_depth=3)
    assert st.camera_rays == 100 * 60 * 16 and float(s.film_to_rgb(xyz, wt).mean()) > 0.2


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "core/src/sobol_matrices.rs")), reason="reference tree not present")
def test_sobol_fixture_matches_reference_tables():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_sobol_fixture.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    src = open(mk.REF).read()
    m32, vdc, vdci = _sobol_tables()
    assert np.array_equal(m32, np.array(mk.table(src, "SOBOL_MATRICES_32")[: len(m32)], np.uint32))
    assert np.array_equal(vdc, np.array(mk.table(src, "VD_C_SOBOL_MATRICES")[: len(vdc)], np.uint64))
    assert np.array_equal(vdci, np.array(mk.table(src, "VD_C_SOBOL_MATRICES_INV")[: len(vdci)], np.uint64))


def test_spot_light_closed_form(host):
    """SpotLight (lights/src/spot.rs): a Lambertian floor lit by one spot, max_depth 1, one centred sample per pixel.  Every pixel must
    equal Kd/pi * I * falloff(cos) / d^2 * cos_surface evaluated in float64 at the camera ray's hit point."""
    import pbrt_hip
    from oracle_binding import OracleScene
    I4 = (np.eye(4, dtype=np.float32).reshape(16),) * 2
    Iv, kd, cone, delta = np.array([20.0, 15.0, 10.0]), np.array([0.6, 0.5, 0.4]), 35.0, 12.0
    res = 48
    with OracleScene() as o:
        l2w, w2l, c_tw, c_fs = host.spot(I4, [0.3, -0.2, 2.0], [0.5, 0.1, 0.0], cone, delta)
        assert c_tw == pytest.approx(np.cos(np.radians(cone)), rel=1e-6) and c_fs == pytest.approx(np.cos(np.radians(cone - delta)), rel=1e-6)
        o.add_light_spot(Iv, l2w, w2l, c_tw, c_fs)
        m = o.add_material_matte(kd, 0.0)
        o.add_mesh(np.float32([[-4, -4, 0], [4, -4, 0], [4, 4, 0], [-4, 4, 0]]), [0, 1, 2, 0, 2, 3], m)
        w2c, c2w = host.look_at([0, -3, 4], [0.2, 0, 0], [0, 0, 1])
        o.set_camera_perspective(host.perspective_raster_to_camera(50.0, res, res), c2w)
        cb, table, sb = host.film_box(res, res)
        o.set_film(res, res, cb, (0.5, 0.5), table)
        o.set_sampler(0, 1, sb, sample_at_pixel_center=True)
        o.build_accel(0, 4)
        xyz, wt, st = o.render_path(max_depth=1, light_strategy=0)
        rgb = o.film_to_rgb(xyz, wt).reshape(res, res, 3).astype(np.float64)
        rays, pfilm = o.generate_camera_rays([0, 0, res, res], 0)
    ro = rays["o"].astype(np.float64); rd = rays["d"].astype(np.float64)
    t = -ro[:, 2] / rd[:, 2]
    P = ro + rd * t[:, None]
    pl = np.array([0.3, -0.2, 2.0]); axis = np.array([0.5, 0.1, 0.0]) - pl; axis /= np.linalg.norm(axis)
    w = P - pl; d2 = (w ** 2).sum(1); wn = w / np.sqrt(d2)[:, None]
    cos_l = wn @ axis
    ctw, cfs = np.cos(np.radians(cone)), np.cos(np.radians(cone - delta))
    dl = np.clip((cos_l - ctw) / (cfs - ctw), 0.0, 1.0)
    fall = np.where(cos_l < ctw, 0.0, np.where(cos_l >= cfs, 1.0, dl ** 4))
    cos_s = np.abs(wn[:, 2])
    expect = (kd / np.pi)[None, :] * Iv[None, :] * (fall / d2 * cos_s)[:, None]
    got = rgb.reshape(-1, 3)
    lit = fall > 1e-3
    assert lit.sum() > 100 and (~lit).sum() > 100
    # XYZ round trip of the film (rgb -> xyz -> rgb) costs ~1e-6 relative; the falloff edge amplifies f32 rounding of cos
    assert np.allclose(got[lit], expect[lit], rtol=2e-3, atol=1e-6)
    assert np.abs(got[fall == 0]).max() < 1e-7


def test_projection_and_goniometric_lights_closed_form(host):
    """ProjectionLight / GonioPhotometricLight (lights/src/projection.rs, goniometric.rs): a Lambertian floor under one such light, max_depth 1, centred
    samples.  Projection: Kd/pi * I * texel / d^2 * cos inside the frustum |x/z| <= aspect tan(fov/2), |y/z| <= tan(fov/2) (fov spans the SHORTER image
    axis), zero outside; a two-texel image tells left from right.  Goniometric: rows of the diagram are selected by the angle from the light's +y axis."""
    import pbrt_hip
    from oracle_binding import OracleScene
    I4 = (np.eye(4, dtype=np.float32).reshape(16),) * 2
    Iv, kd, fov, res = np.array([30.0, 20.0, 10.0]), np.array([0.6, 0.5, 0.4]), 40.0, 40
    pl = np.array([0.2, -0.1, 2.5])
    l2w = host.compose(host.compose(I4, host.translate(pl)), host.rotate(180.0, [1, 0, 0]))      # the light looks down: its +z is world -z, its +y world -y
    two = np.zeros((1, 2, 3), np.float32); two[0, 0] = (1.0, 0.25, 0.0); two[0, 1] = (0.0, 0.25, 1.0)   # aspect 2: left texel reddish, right texel bluish
    rows = np.zeros((4, 1, 3), np.float32); rows[:, 0, 0] = (0.1, 0.4, 0.7, 1.0); rows[:, 0, 1] = 0.5; rows[:, 0, 2] = (1.0, 0.7, 0.4, 0.1)

    def render(add_light):
        with OracleScene() as o:
            add_light(o)
            o.add_mesh(np.float32([[-6, -6, 0], [6, -6, 0], [6, 6, 0], [-6, 6, 0]]), [0, 1, 2, 0, 2, 3], o.add_material_matte(kd, 0.0))
            w2c, c2w = host.look_at([0, -2, 7], [0.1, 0, 0], [0, 0, 1])
            o.set_camera_perspective(host.perspective_raster_to_camera(60.0, res, res), c2w)
            cb, table, sb = host.film_box(res, res)
            o.set_film(res, res, cb, (0.5, 0.5), table)
            o.set_sampler(0, 1, sb, sample_at_pixel_center=True)
            o.build_accel(0, 4)
            xyz, wt, st = o.render_path(max_depth=1, light_strategy=0)
            rgb = o.film_to_rgb(xyz, wt).reshape(-1, 3).astype(np.float64)
            rays, _ = o.generate_camera_rays([0, 0, res, res], 0)
        ro, rd = rays["o"].astype(np.float64), rays["d"].astype(np.float64)
        P = ro + rd * (-ro[:, 2] / rd[:, 2])[:, None]
        w = P - pl; d2 = (w ** 2).sum(1); wn = w / np.sqrt(d2)[:, None]
        base = (kd / np.pi)[None, :] * Iv[None, :] * (np.abs(wn[:, 2]) / d2)[:, None]
        wl = np.stack([w[:, 0], -w[:, 1], -w[:, 2]], axis=1)      # world -> light: rotate 180 degrees about x
        return rgb, base, wl

    tan_h = np.tan(np.radians(fov) / 2)
    rgb, base, wl = render(lambda o: o.add_light_projection(Iv, l2w[0], l2w[1], fov, two))
    px, py = wl[:, 0] / wl[:, 2] / tan_h, wl[:, 1] / wl[:, 2] / tan_h     # screen coordinates: [-2, 2] x [-1, 1]
    inside = (np.abs(px) < 1.98) & (np.abs(py) < 0.98); outside = (np.abs(px) > 2.02) | (np.abs(py) > 1.02)
    assert inside.sum() > 50 and outside.sum() > 50
    assert np.abs(rgb[outside]).max() < 1e-9
    sx = (px + 2) / 4 * 2 - 0.5                                              # MIPMap::triangle: continuous texel coordinate, Repeat wrap
    fx = sx - np.floor(sx); t0 = np.mod(np.floor(sx).astype(int), 2)
    tex = (1 - fx)[:, None] * two[0, t0].astype(np.float64) + fx[:, None] * two[0, 1 - t0].astype(np.float64)
    assert np.allclose(rgb[inside], (base * tex)[inside], rtol=2e-3, atol=1e-7)
    left = inside & (np.abs(px + 1.0) < 0.2); right = inside & (np.abs(px - 1.0) < 0.2)      # around the two texel centres
    assert (rgb[left][:, 0] > 5 * rgb[left][:, 2]).all() and (rgb[right][:, 2] > 2 * rgb[right][:, 0]).all()
    rgb0, base0, _ = render(lambda o: o.add_light_projection(Iv, l2w[0], l2w[1], fov, None))   # no image: white inside a square frustum
    sq = (np.abs(px) < 0.98) & (np.abs(py) < 0.98)
    assert np.allclose(rgb0[sq], base0[sq], rtol=2e-3) and np.abs(rgb0[(np.abs(px) > 1.02) | (np.abs(py) > 1.02)]).max() < 1e-9

    rgb, base, wl = render(lambda o: o.add_light_goniometric(Iv, l2w[0], l2w[1], rows))
    wln = wl / np.linalg.norm(wl, axis=1)[:, None]
    theta = np.arccos(np.clip(wln[:, 1], -1, 1))                               # after swapping y and z the polar axis is the light's +y
    sy = theta / np.pi * 4 - 0.5
    fy = sy - np.floor(sy); r0 = np.mod(np.floor(sy).astype(int), 4); r1 = np.mod(r0 + 1, 4)
    tex = (1 - fy)[:, None] * rows[r0, 0].astype(np.float64) + fy[:, None] * rows[r1, 0].astype(np.float64)
    assert np.allclose(rgb, base * tex, rtol=2e-3, atol=1e-7)
    rgb1, base1, _ = render(lambda o: o.add_light_goniometric(Iv, l2w[0], l2w[1], None))          # no diagram: a point light
    assert np.allclose(rgb1, base1, rtol=2e-3, atol=1e-7)
