"""The oracle-only Sphere (oracle_scene.hpp `Sphere`, restating shapes/src/sphere.rs:10-330 with core/src/efloat.rs) and BASELINE.json's configs[0]
(`scenes/shapes/sphere.pbrt` under the path integrator, maxdepth 4, 16 spp: "CPU reference only (plumbing)").  The product renders triangles only,
so these are CPU tests of the checker: closed forms in float64 pin the restatement, then the configs[0] scene runs end to end."""
import ctypes as C

import numpy as np
import pytest

import pbrt_hip
from oracle_binding import oracle_binding

IDENT = (pbrt_hip.IDENTITY.copy(), pbrt_hip.IDENTITY.copy())


def _fp(a):
    return np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))


def probe(o2w, radius, zmin, zmax, phimax, o, d, t_max=np.inf, flags=0):
    L = oracle_binding().lib
    fp = C.POINTER(C.c_float)
    L.oracle_sphere_probe.argtypes = [fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, fp, fp, C.c_float, fp]
    out = np.zeros(16, np.float32)
    L.oracle_sphere_probe(_fp(o2w[0]), _fp(o2w[1]), radius, zmin, zmax, phimax, flags, _fp(o), _fp(d), t_max, out.ctypes.data_as(fp))
    if out[0] == 0.0:
        return None
    return dict(t=float(out[1]), p=out[2:5].astype(np.float64), n=out[5:8].astype(np.float64), uv=out[8:10].astype(np.float64), p_error=out[10:13].astype(np.float64))


def analytic(radius, zmin, zmax, phimax_deg, o, d, t_max=np.inf):
    """float64 ray / partial-sphere intersection in object space: nearest root whose point passes the clipping tests, as sphere.rs:121-171 walks them"""
    o, d = np.asarray(o, np.float64), np.asarray(d, np.float64)
    a, b, c = d @ d, 2.0 * (d @ o), o @ o - radius * radius
    disc = b * b - 4 * a * c
    if disc < 0:
        return None
    q = -0.5 * (b - np.sqrt(disc)) if b < 0 else -0.5 * (b + np.sqrt(disc))
    t0, t1 = sorted((q / a, c / q))
    zlo, zhi = np.clip(min(zmin, zmax), -radius, radius), np.clip(max(zmin, zmax), -radius, radius)
    pm = np.radians(np.clip(phimax_deg, 0, 360))

    def ok(t):
        p = o + t * d
        phi = np.arctan2(p[1], p[0]); phi = phi + 2 * np.pi if phi < 0 else phi
        return not ((zlo > -radius and p[2] < zlo) or (zhi < radius and p[2] > zhi) or phi > pm), p, phi
    if t0 > t_max or t1 <= 0:
        return None
    cand = [t for t in ((t0, t1) if t0 > 0 else (t1,)) if t <= t_max]
    if not cand or (t0 <= 0 and t1 > t_max):
        return None
    first = t0 if t0 > 0 else t1
    good, p, phi = ok(first)
    if not good:
        if first == t1 or t1 > t_max:
            return None
        good, p, phi = ok(t1); first = t1
        if not good:
            return None
    th = np.arccos(np.clip(p[2] / radius, -1, 1))
    tmin, tmax = np.arccos(np.clip(zlo / radius, -1, 1)), np.arccos(np.clip(zhi / radius, -1, 1))
    return dict(t=first, p=p, uv=np.array([phi / pm, (th - tmin) / (tmax - tmin)]))


def test_full_sphere_hits_match_float64():
    rng = np.random.default_rng(5)
    n_hit = 0
    for _ in range(400):
        o = rng.normal(size=3); o = o / np.linalg.norm(o) * rng.uniform(1.5, 6.0)
        target = rng.uniform(-1.2, 1.2, 3)
        d = (target - o) * rng.uniform(0.3, 3.0)  # unnormalised directions, as the integrator hands them over
        got, want = probe(IDENT, 1.0, -1.0, 1.0, 360.0, o, d), analytic(1.0, -1.0, 1.0, 360.0, o.astype(np.float32), d.astype(np.float32))
        if want is None or got is None:
            if (want is None) != (got is None):  # grazing rays may differ: the discriminant must then be tiny
                of, df = o.astype(np.float32).astype(np.float64), d.astype(np.float32).astype(np.float64)
                disc = (2 * df @ of) ** 2 - 4 * (df @ df) * (of @ of - 1)
                assert abs(disc) < 1e-3 * (df @ df), (o, d)
            continue
        n_hit += 1
        assert abs(got["t"] - want["t"]) <= 2e-5 * max(1.0, want["t"])
        assert np.allclose(got["p"], want["p"], atol=2e-5)
        assert abs(np.linalg.norm(got["p"]) - 1.0) < 1e-6           # the refined point lies on the sphere
        assert np.allclose(got["n"], got["p"], atol=1e-5)           # outward normal = p / r for an un-reversed, right-handed sphere
        assert np.allclose(got["uv"], want["uv"], atol=2e-5)
        assert np.all(got["p_error"] >= 0) and np.all(got["p_error"] < 1e-5)
    assert n_hit > 150


def test_inside_origin_behind_and_tmax():
    assert probe(IDENT, 1.0, -1, 1, 360, (0, 0, 0), (0, 0, 1))["t"] == pytest.approx(1.0, abs=1e-6)         # from inside: the far root
    assert probe(IDENT, 1.0, -1, 1, 360, (0, 0, 3), (0, 0, 1)) is None                                     # sphere behind the ray
    assert probe(IDENT, 1.0, -1, 1, 360, (0, 0, 3), (0, 0, -1), t_max=1.5) is None                         # t_max in front of the sphere
    h = probe(IDENT, 1.0, -1, 1, 360, (0, 0, 3), (0, 0, -1))
    assert h["t"] == pytest.approx(2.0, abs=1e-6) and h["p"][0] == pytest.approx(1e-5, rel=1e-3)          # the pole: p.x = 1e-5 * radius (sphere.rs:137-139)
    assert probe(IDENT, 1.0, -1, 1, 360, (0, 0, 0.5), (0, 0, -1), t_max=1.0) is None                       # inside, far root beyond t_max
    assert probe(IDENT, 2.0, -2, 2, 360, (5, 0, 0), (-1, 0, 0))["t"] == pytest.approx(3.0, abs=1e-6)


def test_partial_sphere_clipping_matches_float64():
    rng = np.random.default_rng(9)
    kinds = 0
    for _ in range(600):
        zmin, zmax, phimax = rng.uniform(-1, 0.2), rng.uniform(0.3, 1.0), rng.uniform(40, 330)
        o = rng.normal(size=3); o = o / np.linalg.norm(o) * rng.uniform(1.5, 4.0)
        d = rng.uniform(-0.9, 0.9, 3) - o
        got = probe(IDENT, 1.0, zmin, zmax, phimax, o, d)
        want = analytic(1.0, np.float32(zmin), np.float32(zmax), np.float32(phimax), o.astype(np.float32), d.astype(np.float32))
        if (got is None) != (want is None):
            # a root within rounding of a clipping plane / of phi_max may fall on the other side in f32
            full = analytic(1.0, -1, 1, 360, o.astype(np.float32), d.astype(np.float32))
            assert full is not None
            continue
        if got is None:
            continue
        kinds += 1
        assert abs(got["t"] - want["t"]) <= 3e-5 * max(1.0, want["t"])
        assert np.allclose(got["uv"], want["uv"], atol=1e-4)
        assert -1e-6 <= got["uv"][0] <= 1 + 1e-6 and -1e-5 <= got["uv"][1] <= 1 + 1e-5
        assert np.float32(zmin) - 1e-6 <= got["p"][2] <= np.float32(zmax) + 1e-6
    assert kinds > 100


def test_transformed_sphere_and_orientation():
    host = pbrt_hip.Host()
    t = host.compose(host.compose(host.translate((1.0, -2.0, 0.5)), host.rotate(35.0, (0.3, 1.0, 0.2))), host.scale((1.5, 1.5, 1.5)))
    o, d = np.array([6.0, 3.0, 2.0]), np.array([-5.2, -4.9, -1.4])
    got = probe(t, 1.0, -1, 1, 360, o, d)
    m = np.asarray(t[0], np.float64).reshape(4, 4); mi = np.asarray(t[1], np.float64).reshape(4, 4)
    oo, do = (mi @ np.append(o, 1))[:3], mi[:3, :3] @ d
    want = analytic(1.0, -1, 1, 360, oo, do)
    assert abs(got["t"] - want["t"]) < 1e-5
    pw = (m @ np.append(want["p"], 1))[:3]
    assert np.allclose(got["p"], pw, atol=2e-5)
    centre = m[:3, 3]
    assert np.allclose(got["n"], (pw - centre) / np.linalg.norm(pw - centre), atol=1e-5)
    assert got["p_error"].max() < 2e-5 and got["p_error"].min() > 0
    # ReverseOrientation flips the normal; a mirroring transform (negative determinant) flips it back (surface_interaction.rs:72-77)
    rev = probe(t, 1.0, -1, 1, 360, o, d, flags=1)
    assert np.allclose(rev["n"], -got["n"], atol=1e-6)
    mir = host.compose(t, host.scale((1.0, 1.0, -1.0)))
    a, b = probe(mir, 1.0, -1, 1, 360, o, d), probe(mir, 1.0, -1, 1, 360, o, d, flags=1)
    c = (np.asarray(mir[0], np.float64).reshape(4, 4))[:3, 3]
    outward = (a["p"] - c) / np.linalg.norm(a["p"] - c)
    assert np.allclose(a["n"], -outward, atol=1e-5) and np.allclose(b["n"], outward, atol=1e-5)


def add_sphere(scene, t, radius=1.0, zmin=None, zmax=None, phimax=360.0, material=0, reverse=False):
    L = scene.b.lib
    fp = C.POINTER(C.c_float)
    L.oracle_add_sphere.argtypes = [C.c_void_p, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32]
    zmin = -radius if zmin is None else zmin
    zmax = radius if zmax is None else zmax
    scene._chk(L.oracle_add_sphere(scene.h, _fp(t[0]), _fp(t[1]), radius, zmin, zmax, phimax, material, 1 if reverse else 0))


def _camera(scene, host, eye, look, up, fov, xres, yres, spp):
    w2c, c2w = host.look_at(eye, look, up)
    scene.set_camera_perspective(host.perspective_raster_to_camera(fov, xres, yres), c2w)
    cb, table, sb = host.film_box(xres, yres)
    scene.set_film(xres, yres, cb, (0.5, 0.5), table)
    scene.set_sampler(0, spp, sb)


def test_white_furnace_on_a_sphere():
    """A convex matte sphere of albedo Kd inside a constant environment L: every bounce ray escapes, emission is only added for camera and specular rays,
    so the estimator's expectation is Kd * L at every pixel of the sphere (SURVEY §8c's furnace test) — and exactly L beside it."""
    host = pbrt_hip.Host()
    with pbrt_hip.Scene(oracle_binding()) as s:
        m = s.add_material_matte((0.6, 0.6, 0.6))
        s.add_light_infinite((1.0, 1.0, 1.0))
        add_sphere(s, IDENT, 1.0, material=m)
        _camera(s, host, (0, -4, 0), (0, 0, 0), (0, 0, 1), 35.0, 48, 48, 64)
        s.build_accel(0, 4)
        xyz, wt, st = s.render_path(max_depth=5)
        rgb = s.film_to_rgb(xyz, wt)
    centre = rgb[16:32, 16:32]
    assert abs(centre.mean() - 0.6) < 0.01
    assert np.allclose(rgb[0, 0], 1.0, atol=1e-6) and np.allclose(rgb[47, 47], 1.0, atol=1e-6)
    assert st.shadow_rays > 0 and st.regular_rays > 48 * 48 * 64


def test_sphere_occludes_and_is_occluded_like_its_triangulation():
    """A sphere and a finely triangulated sphere of the same radius give the same image up to the faceting: the Sphere sits in the same BVH, shadow rays see it,
    the ground plane below receives its shadow."""
    host = pbrt_hip.Host()

    def render(as_sphere):
        with pbrt_hip.Scene(oracle_binding()) as s:
            m = s.add_material_matte((0.5, 0.5, 0.5))
            s.add_light_distant((3.0, 3.0, 3.0), (0.0, 0.0, 1.0))
            ground = np.array([[-6, -6, -1], [6, -6, -1], [6, 6, -1], [-6, 6, -1]], np.float32)
            s.add_mesh(ground, np.array([0, 1, 2, 0, 2, 3], np.uint32), m)
            if as_sphere:
                add_sphere(s, IDENT, 1.0, material=m)
            else:
                nu, nv = 96, 48
                u, v = np.meshgrid(np.linspace(0, 2 * np.pi, nu + 1), np.linspace(0, np.pi, nv + 1))
                P = np.stack([np.sin(v) * np.cos(u), np.sin(v) * np.sin(u), np.cos(v)], -1).reshape(-1, 3).astype(np.float32)
                idx = []
                for j in range(nv):
                    for i in range(nu):
                        a = j * (nu + 1) + i
                        idx += [a, a + nu + 1, a + 1, a + 1, a + nu + 1, a + nu + 2]
                s.add_mesh(P, np.array(idx, np.uint32), m)
            _camera(s, host, (0, -7, 4), (0, 0, -0.5), (0, 0, 1), 40.0, 64, 64, 16)
            s.build_accel(0, 4)
            xyz, wt, _ = s.render_path(max_depth=3)
            return s.film_to_rgb(xyz, wt)
    a, b = render(True), render(False)
    assert np.abs(a - b).mean() < 0.01 * b.mean() + 1e-3
    assert np.mean(np.abs(a - b) > 0.1) < 0.03  # silhouette and shadow-edge pixels only
    assert a.min() == 0.0 and a.max() > 0.5    # a shadow exists, lit ground exists


def configs0_scene(s, host, xres, yres, spp):
    """scenes/shapes/sphere.pbrt with the substitutions SURVEY §8d prescribes: PathIntegrator maxdepth 4, `pixelsamples` spp, and a constant Kd in place of
    the uv-grid image map (its PNG lives outside the repository)."""
    s.add_light_infinite((1.2, 1.2, 1.1))
    m1 = s.add_material_matte((0.5, 0.5, 0.5))

    def ctm(*steps):
        t = IDENT
        for st_ in steps:
            t = host.compose(t, st_)
        return t
    add_sphere(s, ctm(host.translate((-1.75, 0, 0)), host.scale((1.5, 1.5, 1.5)), host.rotate(135, (1, 0, 0)), host.rotate(-15, (0, 0, 1)), host.rotate(15, (0, 1, 0))),
               1.0, material=m1)
    add_sphere(s, ctm(host.translate((1.75, 0, 0)), host.scale((1.5, 1.5, 1.5)), host.rotate(-100, (1, 0, 0)), host.rotate(-90, (0, 1, 0)), host.rotate(-30, (0, 0, 1)),
                      host.rotate(-20, (1, 0, 0))), 1.0, zmin=-1.0, zmax=0.0, phimax=210.0, material=m1)
    m2 = s.add_material_matte((0.5, 0.5, 0.5))
    t = host.translate((0, 0, -1.5))
    P = host.transform_points(t[0], np.array([[-20, -20, 0], [20, -20, 0], [20, 20, 0], [-20, 20, 0]], np.float32))
    s.add_mesh(P, np.array([0, 1, 2, 0, 2, 3], np.uint32), m2, UV=np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32))
    _camera(s, host, (0, 5, 1.5), (0, 0, 0), (0, 0, 1), 45.0, xres, yres, spp)
    s.build_accel(0, 4)


def test_configs0_sphere_scene_runs_in_the_oracle():
    """BASELINE.json configs[0]: 800 x 400 in the reference; the test renders the same scene at 200 x 100 (every code path, a sixteenth of the pixels)
    and a 80 x 40 crop's worth twice, to show the run is deterministic."""
    host = pbrt_hip.Host()

    def run(xres, yres):
        with pbrt_hip.Scene(oracle_binding()) as s:
            configs0_scene(s, host, xres, yres, 16)
            xyz, wt, st = s.render_path(max_depth=4)
            return s.film_to_rgb(xyz, wt), st.as_dict()
    rgb, st = run(200, 100)
    assert np.all(np.isfinite(rgb)) and rgb.min() >= 0.0
    assert st["camera_rays"] == 200 * 100 * 16 and st["regular_rays"] > st["camera_rays"] and st["shadow_rays"] > 0
    sky = rgb[2, 100]
    assert np.allclose(sky, (1.2, 1.2, 1.1), atol=1e-5)            # above the horizon: the environment itself
    left, right = rgb[45:60, 55:70], rgb[45:60, 130:145]            # on the two spheres
    assert 0.2 < left.mean() < 0.75 and 0.1 < right.mean() < 0.75   # lit matte surfaces, darker than the sky
    ground = rgb[90:98, 80:120]
    assert 0.2 < ground.mean() < 0.7
    # the cut-away sphere shows its inside: its silhouette is smaller than the full one's
    is_obj = lambda block: np.mean(np.abs(block - np.array([1.2, 1.2, 1.1])).max(-1) > 0.05)
    assert is_obj(rgb[20:70, 100:180]) < is_obj(rgb[20:70, 20:100])
    a, _ = run(80, 40); b, _ = run(80, 40)
    assert np.array_equal(a, b)


def test_configs0_silhouettes_equal_the_references_own_render():
    """The reference ships its render of this scene (renders/shapes/sphere.png, 800 x 400); tests/golden/sphere_sky_mask.npz is the set of its pixels that show the
    pure environment (made by tests/golden/make_sphere_mask.py).  The oracle's render of the same file must show sky / not-sky in exactly the same pixels,
    up to the one-pixel band along the silhouettes where partial coverage decides: this pins LookAt + perspective, the five-deep CTM of each sphere, the Sphere's
    quadratic and its zmin / zmax / phimax cut, and the ground quad against output of the reference itself."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sphere_sky_mask.npz"))
    h, w = (int(v) for v in z["shape"])
    sky = np.unpackbits(z["sky"])[: h * w].reshape(h, w).astype(bool)
    host = pbrt_hip.Host()
    with pbrt_hip.Scene(oracle_binding()) as s:
        configs0_scene(s, host, w, h, 4)
        xyz, wt, _ = s.render_path(max_depth=1)
        rgb = s.film_to_rgb(xyz, wt)
    mine = np.all(np.abs(rgb - np.array([1.2, 1.2, 1.1], np.float32)) < 1e-4, -1)

    def interior(m):
        p = np.pad(m, 1, mode="edge"); out = np.ones_like(m)
        for dy in range(3):
            for dx in range(3):
                out &= p[dy:dy + h, dx:dx + w]
        return out
    in_sky, in_obj = interior(sky), interior(~sky)
    assert in_sky.sum() + in_obj.sum() > 0.975 * h * w      # the band is thin
    assert not np.any(in_sky & ~mine) and not np.any(in_obj & mine)
    assert (sky != mine).mean() < 0.006
