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
    ok = (a["prim"] == b["prim"])
    for f in ("t", "b0", "b1", "b2"):
        ok &= (a[f].view(np.uint32) == b[f].view(np.uint32))
    return ok
