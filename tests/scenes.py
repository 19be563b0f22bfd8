"""Shared scene builders for the tests: every scene is captured through the same pbrt_hip.Scene calls into either the
product binding or the oracle binding."""
import numpy as np

import pbrt_hip


def random_rays(n, seed, bound=1.2, t_max=np.inf):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, pbrt_hip.RAY_DTYPE)
    rays["o"] = rng.uniform(-bound * 2, bound * 2, (n, 3)).astype(np.float32)
    target = rng.uniform(-bound, bound, (n, 3)).astype(np.float32)
    d = target - rays["o"]
    # half the rays normalised, half not (shadow rays are un-normalised in the reference)
    nrm = np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d[: n // 2] = d[: n // 2] / nrm[: n // 2]
    rays["d"] = d
    rays["t_max"] = t_max
    rays["t_max"][n // 2:] = np.float32(1.0 - 1e-4) if not np.isfinite(t_max) else t_max
    rays["time"] = 0.0
    return rays


def axis_rays():
    """Degenerate directions: zero components (inv_dir = +-inf, NaN slabs), axis-aligned, negative zero."""
    dirs = []
    for ax in range(3):
        for sgn in (1.0, -1.0):
            d = [0.0, 0.0, 0.0]; d[ax] = sgn; dirs.append(d)
            d = [-0.0, -0.0, -0.0]; d[ax] = sgn; dirs.append(d)
    dirs += [[1, 1, 0], [0, 1, 1], [1, 0, -1], [1e-30, 1, 1e-30], [1, 1e-38, 0]]
    origins = [[0, 0, 0], [0.1, -3, 0.2], [-3, 0.05, 0.0], [0.0, 0.0, 3.0], [2, 2, 2]]
    rays = np.zeros(len(dirs) * len(origins), pbrt_hip.RAY_DTYPE)
    k = 0
    for o in origins:
        for d in dirs:
            rays[k]["o"] = o; rays[k]["d"] = d; rays[k]["t_max"] = np.inf; k += 1
    return rays


def grid_mesh(n=8, z=0.0, size=1.0):
    """(n x n) quad grid -> shared vertices, exact edge/vertex hits exercise the f64 fallback and t ties."""
    xs = np.linspace(-size, size, n + 1, dtype=np.float32)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    P = np.stack([X.ravel(), Y.ravel(), np.full(X.size, z, np.float32)], axis=1).astype(np.float32)
    idx = []
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i; b = a + 1; c = a + n + 1; d = c + 1
            idx += [a, b, d, a, d, c]
    return P, np.array(idx, np.uint32)


def build_pair(capture, oracle_scene_cls):
    """capture(scene) is applied to a product Scene and to an OracleScene."""
    prod = pbrt_hip.Scene()
    orc = oracle_scene_cls()
    capture(prod); capture(orc)
    return prod, orc


def hits_equal(a, b):
    """Bit-exact comparison of (t, prim, b0, b1, b2)."""
    ok = (a["prim"] == b["prim"]) & (a["pad"][:, 1] == b["pad"][:, 1])   # pad[1]: instance number + 1
    for f in ("t", "b0", "b1", "b2"):
        ok &= (a[f].view(np.uint32) == b[f].view(np.uint32))
    return ok


def cornell_like(host, with_normals=False, with_uv=False, two_sided=False, sigma=0.0, reverse=False):
    """A closed-ish box with an emissive quad on the ceiling + a tilted inner quad: exercises DiffuseAreaLight sampling /
    pdf_solid_angle / Le on hit, Oren-Nayar, optional per-vertex N / UV shading frames and reverse_orientation.
    Returns a capture(scene) closure."""
    def quad(a, b, c, d):
        P = np.array([a, b, c, d], np.float32)
        return P, np.array([0, 1, 2, 0, 2, 3], np.uint32)

    walls = [
        quad([-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1]),      # floor
        quad([-1, -1, 1], [-1, 1, 1], [1, 1, 1], [1, -1, 1]),          # ceiling
        quad([-1, 1, -1], [1, 1, -1], [1, 1, 1], [-1, 1, 1]),          # back
        quad([-1, -1, -1], [-1, 1, -1], [-1, 1, 1], [-1, -1, 1]),      # left
        quad([1, -1, -1], [1, -1, 1], [1, 1, 1], [1, 1, -1]),          # right
    ]
    light = quad([-0.3, -0.3, 0.98], [0.3, -0.3, 0.98], [0.3, 0.3, 0.98], [-0.3, 0.3, 0.98])
    inner = quad([-0.5, 0.2, -0.6], [0.4, 0.0, -0.7], [0.5, 0.3, 0.1], [-0.4, 0.5, 0.2])

    def capture(s):
        white = s.add_material_matte((0.7, 0.7, 0.7), sigma)
        red = s.add_material_matte((0.6, 0.1, 0.1), 0.0)
        for k, (P, idx) in enumerate(walls):
            s.add_mesh(P, idx, red if k == 3 else white)
        lid = s.add_light_diffuse_area((8.0, 7.0, 6.0), 2, two_sided=two_sided)
        # the light faces down (-z) only if its winding gives n = -z; flip with reverse_orientation otherwise
        s.add_mesh(light[0], light[1], white, first_area_light=lid, reverse_orientation=True)
        P, idx = inner
        N = UV = None
        if with_normals:
            n = np.cross(P[1] - P[0], P[2] - P[0]); n /= np.linalg.norm(n)
            N = np.tile(n, (4, 1)).astype(np.float32) + np.array([[0.1, 0, 0], [0, 0.1, 0], [-0.1, 0, 0], [0, -0.1, 0]], np.float32)
        if with_uv:
            UV = np.array([[0, 0], [2, 0.1], [2.2, 1.5], [0.1, 1.4]], np.float32)
        s.add_mesh(P, idx, white, N=N, UV=UV, reverse_orientation=reverse)
        w2c, c2w = host.look_at([0, -3.4, 0], [0, 0, 0], [0, 0, 1])
        s.set_camera_perspective(host.perspective_raster_to_camera(40.0, 48, 48), c2w)
        cb, table, sb = host.film_box(48, 48)
        s.set_film(48, 48, cb, (0.5, 0.5), table)
        s.set_sampler(0, 8, sb)
        s.build_accel(0, 4)
    return capture
