"""-m gpu: K2/K3/K5 (BVH traversal + watertight triangle test) against the CPU oracle, through the C ABI.
Bar: bit-exact (t, prim, b0, b1, b2) and occlusion flags."""
import numpy as np
import pytest

import pbrt_hip
import scenes
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu


def _pair(n_tris, seed, host, extra=None):
    P, idx = host.gen_random_tris(n_tris, seed)

    def capture(s):
        m = s.add_material_matte()
        s.add_mesh(P, idx, m)
        if extra:
            extra(s, m)
        s.build_accel(0, 4)
    return scenes.build_pair(capture, OracleScene)


@pytest.mark.parametrize("n_tris,seed", [(1, 1), (2, 2), (5, 3), (1000, 1), (100000, 2)])
def test_closest_hit_random_rays_bit_exact(host, n_tris, seed):
    prod, orc = _pair(n_tris, seed, host)
    rays = np.concatenate([scenes.random_rays(200000, seed + 10), scenes.axis_rays()])
    got = prod.intersect_batch(rays)
    want, st = orc.intersect_batch_stats(rays)
    eq = scenes.hits_equal(got, want)
    assert eq.all(), f"{(~eq).sum()} of {len(rays)} rays differ; first: {np.flatnonzero(~eq)[:5]} got {got[~eq][:3]} want {want[~eq][:3]}"
    assert (want["prim"] != 0xFFFFFFFF).sum() > 0 or n_tris < 10


@pytest.mark.parametrize("n_tris,seed", [(1, 1), (5, 3), (1000, 1), (100000, 2)])
def test_any_hit_random_rays_bit_exact(host, n_tris, seed):
    prod, orc = _pair(n_tris, seed, host)
    rays = np.concatenate([scenes.random_rays(200000, seed + 20), scenes.axis_rays()])
    got = prod.occluded_batch(rays)
    want, _ = orc.occluded_batch_stats(rays)
    assert np.array_equal(got, want), f"{(got != want).sum()} differ"


def test_shared_vertex_grid_edges_and_ties(host):
    """Rays aimed exactly at shared vertices/edges: f64 fallback (triangle.rs:483-495) and equal-t ties
    (last tested triangle wins, triangle.rs:512-516) must resolve exactly like the reference order."""
    P, idx = scenes.grid_mesh(8)

    def capture(s):
        m = s.add_material_matte()
        s.add_mesh(P, idx, m)
        s.build_accel(0, 4)
    prod, orc = scenes.build_pair(capture, OracleScene)
    n = len(P)
    rays = np.zeros(3 * n, pbrt_hip.RAY_DTYPE)
    rays["t_max"] = np.inf
    rays["o"][:n] = P + np.array([0, 0, 2], np.float32); rays["d"][:n] = [0, 0, -1]                     # straight down on vertices
    rays["o"][n:2 * n] = [0.3, -0.2, 3.0]; rays["d"][n:2 * n] = P - np.array([0.3, -0.2, 3.0], np.float32)  # oblique, through vertices
    mid = (P + np.roll(P, 1, axis=0)) * np.float32(0.5)
    rays["o"][2 * n:] = mid + np.array([0, 0, 1], np.float32); rays["d"][2 * n:] = [0, 0, -1]         # edge midpoints
    got = prod.intersect_batch(rays); want, _ = orc.intersect_batch_stats(rays)
    eq = scenes.hits_equal(got, want)
    assert eq.all(), f"{(~eq).sum()} differ"
    assert (want["prim"] != 0xFFFFFFFF).sum() > n


def test_degenerate_and_alpha_triangles(host):
    """Zero-area triangles are rejected after the t test (triangle.rs:567-570); alpha == 0 meshes are invisible to
    intersect, shadowalpha == 0 additionally to intersect_p (triangle.rs:603, 886-893)."""
    P = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0],      # regular
                  [-1, -1, 1], [1, 1, 1], [0, 0, 1],       # collinear -> bogus
                  [-1, -1, 2], [1, -1, 2], [1, 1, 2],      # alpha 0 mesh
                  [-1, -1, 3], [1, -1, 3], [1, 1, 3]], np.float32)

    def capture(s):
        m = s.add_material_matte()
        s.add_mesh(P[0:6], [0, 1, 2, 3, 4, 5], m)
        s.add_mesh(P[6:9], [0, 1, 2], m, alpha=0.0)
        s.add_mesh(P[9:12], [0, 1, 2], m, shadow_alpha=0.0)
        s.build_accel(0, 4)
    prod, orc = scenes.build_pair(capture, OracleScene)
    rays = np.zeros(4, pbrt_hip.RAY_DTYPE)
    rays["o"] = [[0.5, -0.5, 5], [0.25, 0.25, 5], [0.5, -0.5, -5], [0.5, -0.5, 2.5]]
    rays["d"] = [[0, 0, -1], [0, 0, -1], [0, 0, 1], [0, 0, 1]]
    rays["t_max"] = [np.inf, np.inf, np.inf, 1.0]
    got = prod.intersect_batch(rays); want, _ = orc.intersect_batch_stats(rays)
    assert scenes.hits_equal(got, want).all()
    assert want["prim"][0] == 3 and want["prim"][2] == 0   # shadowalpha mesh is visible to closest-hit; alpha-0 mesh is not
    g2 = prod.occluded_batch(rays); w2, _ = orc.occluded_batch_stats(rays)
    assert np.array_equal(g2, w2)
    assert w2[3] == 0  # the only triangle in range has shadowalpha 0


def test_recorded_path_rays_bit_exact(host):
    """Rays a real render traces (primary + bounce + MIS + shadow), recorded by the oracle."""
    spec = pbrt_hip.SceneSpec(n_tris=20000, seed=4, xres=48, yres=48, spp=2)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    pbrt_hip.capture_spec(spec, orc, host); pbrt_hip.capture_spec(spec, prod, host)
    orc.record_rays(1 << 20)
    orc.render_path_ex()
    reg, sh = orc.recorded_rays(False), orc.recorded_rays(True)
    assert len(reg) > 4000 and len(sh) > 1000
    got = prod.intersect_batch(reg); want, _ = orc.intersect_batch_stats(reg)
    assert scenes.hits_equal(got, want).all()
    assert np.array_equal(prod.occluded_batch(sh), orc.occluded_batch_stats(sh)[0])


def test_empty_and_error_paths(host):
    s = pbrt_hip.Scene()
    with pytest.raises(pbrt_hip.PbrtHipError) as e:
        s.intersect_batch(np.zeros(4, pbrt_hip.RAY_DTYPE))
    assert e.value.code == pbrt_hip.ERR_STATE
    m = s.add_material_matte()
    s.build_accel(0, 4)  # empty scene: every ray misses, like BVHAccel with no nodes (bvh/mod.rs:175)
    rays = scenes.random_rays(100, 1)
    h = s.intersect_batch(rays)
    assert (h["prim"] == 0xFFFFFFFF).all() and np.array_equal(h["t"].view(np.uint32), rays["t_max"].view(np.uint32))
    assert not s.occluded_batch(rays).any()
    assert len(s.intersect_batch(np.zeros(0, pbrt_hip.RAY_DTYPE))) == 0
    with pytest.raises(pbrt_hip.PbrtHipError):
        s.add_mesh(np.zeros((3, 3), np.float32), [0, 1, 5], m)  # out-of-bounds index (triangle.rs:252-261)


@pytest.mark.parametrize("split_method,max_prims", [(1, 4), (1, 1), (3, 4)])
def test_other_split_methods_bit_exact(host, split_method, max_prims):
    """HLBVH (hlbvh.rs, incl. its Morton-bit-pattern quirk) and EqualCounts trees on the device: same closest hits and occlusion
    flags as the oracle traversing the oracle's own tree of that kind; the grid mesh puts equal-t ties on shared edges, which
    resolve by traversal order and therefore test the topology."""
    Pg, ig = scenes.grid_mesh(6, z=0.1, size=0.9)
    P, idx = host.gen_random_tris(3000, 21)

    def capture(s):
        m = s.add_material_matte()
        s.add_mesh(P, idx, m)
        s.add_mesh(Pg, ig, m)
        s.build_accel(split_method, max_prims)
    prod, orc = scenes.build_pair(capture, OracleScene)
    n = len(Pg)
    edge = np.zeros(n, pbrt_hip.RAY_DTYPE)
    edge["t_max"] = np.inf; edge["o"] = Pg + np.array([0, 0, 2], np.float32); edge["d"] = [0, 0, -1]
    rays = np.concatenate([scenes.random_rays(30000, 77), scenes.axis_rays(), edge])
    got = prod.intersect_batch(rays)
    want, st = orc.intersect_batch_stats(rays)
    eq = scenes.hits_equal(got, want)
    assert eq.all(), f"{(~eq).sum()} of {len(rays)} rays differ"
    assert np.array_equal(prod.occluded_batch(rays), orc.occluded_batch_stats(rays)[0])
    if split_method == 1:
        assert st.nodes_visited / st.rays > 300   # the reference's HLBVH is not spatially coherent (quirk B10, HISTORY §3a)
