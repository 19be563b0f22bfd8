"""-m gpu: the trees built on the device — HLBVH (csrc/bvh_device.hip: hlbvh.rs + morton.rs as kernels) and SAH, the reference's default (csrc/bvh_sah_device.hip: sah.rs one
tree level per round of kernels) — against the host builder and the oracle.  Same leaves in the same depth-first order, same leaf ends, the same child boxes, the same root
bound (for SAH: the same node array, entry by entry) — so hits cannot depend on where the tree was built."""
import ctypes as C

import numpy as np
import pytest

import pbrt_hip
import scenes
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu


def _build(lib, which, P, idx, n_tris, max_prims, split=1):
    order = np.zeros(n_tris, np.uint32); last = np.zeros(n_tris, np.uint32)
    nodes = np.zeros((max(n_tris - 1, 1), 16), np.uint32); info = np.zeros(5, np.uint64); rb = np.zeros(6, np.float32)
    if which == "host":
        lib.pbrt_hip_host_build_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = lib.pbrt_hip_host_build_bvh(P.ctypes.data, idx.ctypes.data, n_tris, split, max_prims, 4, order.ctypes.data, last.ctypes.data, nodes.ctypes.data, info.ctypes.data, rb.ctypes.data)
    else:
        lib.pbrt_hip_device_build_bvh.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = lib.pbrt_hip_device_build_bvh(0, P.ctypes.data, idx.ctypes.data, n_tris, split, max_prims, order.ctypes.data, last.ctypes.data, nodes.ctypes.data, info.ctypes.data, rb.ctypes.data, None)
    assert rc == 0, rc
    f = nodes.view(np.float32)
    n_int = int(info[0])
    boxes = {(f[k, c], f[k, c + 2], f[k, c + 4], f[k, c + 1], f[k, c + 3], f[k, c + 5]) for k in range(n_int) for c in (0, 6)}
    ends = np.flatnonzero(last); starts = np.concatenate([[0], ends[:-1] + 1]) if len(ends) else np.zeros(0, int)
    leaves = [tuple(order[a:b + 1]) for a, b in zip(starts, ends)]
    return dict(order=order, last=last, info=info[:3].copy(), info5=info.copy(), rb=rb, boxes=boxes, leaves=leaves, nodes=nodes[:n_int].copy())


@pytest.mark.parametrize("n_tris,seed,max_prims", [(1, 1, 4), (2, 1, 4), (3, 5, 4), (17, 2, 4), (300, 8, 2), (5000, 3, 4), (5000, 4, 1), (20000, 6, 8), (300000, 7, 4)])
def test_device_hlbvh_equals_host_and_oracle(host, product, n_tris, seed, max_prims):
    P, idx = host.gen_random_tris(n_tris, seed)
    h = _build(product.lib, "host", P, idx, n_tris, max_prims)
    d = _build(product.lib, "device", P, idx, n_tris, max_prims)
    assert np.array_equal(d["order"], h["order"]), "primitive order differs"
    assert np.array_equal(d["last"], h["last"]), "leaf ends differ"
    assert np.array_equal(d["info"], h["info"])          # interior nodes, leaves, largest leaf
    assert np.array_equal(d["rb"], h["rb"])
    assert d["boxes"] == h["boxes"]
    if n_tris <= 20000:
        orc = OracleScene()
        m = orc.add_material_matte(); orc.add_mesh(P, idx, m); orc.build_accel(1, max_prims)
        onodes = orc.bvh_nodes()
        oprims = np.zeros(n_tris, np.uint32); orc.b.lib.oracle_bvh_ordered_prims(orc.h, oprims.ctypes.data)
        ol = onodes[onodes["n_primitives"] > 0]
        assert d["leaves"] == [tuple(oprims[l["offset"]:l["offset"] + l["n_primitives"]]) for l in ol]
        if len(onodes) > 1:
            assert d["boxes"] == {tuple(n["pmin"]) + tuple(n["pmax"]) for n in onodes[1:]}


def _awkward_tris(n_tris, seed, mode):
    """Triangle soups the SAH build's special cases meet: mode 1 — centroids on a coarse flat lattice (many coincident centroids, empty buckets, ranges that stay leaves
    above max_prims); mode 2 — three quarters of the triangles are one and the same triangle."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n_tris, 3)).astype(np.float32)
    d = (rng.uniform(-1, 1, (n_tris, 3, 3)) * 0.05).astype(np.float32)
    if mode == 1:
        c = np.floor(c * 3).astype(np.float32); c[:, 2] = 0
        d[:] = 0; d[:, 1, 0] = 0.5; d[:, 2, 1] = 0.5
    if mode == 2:
        same = (np.arange(n_tris) & 3) != 0
        c[same] = (0.25, -0.5, 0.75); d[same] = 0; d[same, 1, 0] = 0.05; d[same, 2, 1] = 0.05
    if mode == 3:   # a regular k x k grid of quads in a plane, shared vertices: rows of equal centroid coordinates, bucket boundaries that fall on them, a degenerate axis
        k = int(np.sqrt(n_tris // 2))
        u = np.linspace(-1, 1, k + 1, dtype=np.float32)
        X, Y = np.meshgrid(u, u, indexing="xy")
        P = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size, np.float32)], 1).astype(np.float32)
        i, j = np.meshgrid(np.arange(k), np.arange(k), indexing="xy")
        a = (j * (k + 1) + i).ravel(); b = a + 1; c2 = a + k + 1; d2 = c2 + 1
        idx = np.stack([a, b, d2, a, d2, c2], 1).reshape(-1).astype(np.uint32)
        return np.ascontiguousarray(P), idx
    P = (c[:, None, :] + d).reshape(-1, 3).astype(np.float32)
    return np.ascontiguousarray(P), np.arange(3 * n_tris, dtype=np.uint32)


@pytest.mark.parametrize("n_tris,seed,max_prims,mode", [(1, 1, 4, 0), (2, 1, 4, 0), (3, 5, 4, 0), (5, 2, 1, 0), (17, 2, 4, 0), (300, 8, 2, 0), (5000, 3, 4, 0), (5000, 4, 1, 0), (20000, 6, 8, 0),
                                                        (20000, 9, 255, 0), (3000, 1, 4, 1), (3000, 2, 4, 2), (40000, 3, 4, 1), (300000, 7, 4, 0), (1000000, 11, 4, 0), (20000, 1, 4, 3), (2000000, 1, 4, 3)])
def test_device_sah_build_equals_host_and_oracle(host, product, n_tris, seed, max_prims, mode):
    P, idx = host.gen_random_tris(n_tris, seed) if mode == 0 else _awkward_tris(n_tris, seed, mode)
    n_tris = len(idx) // 3
    h = _build(product.lib, "host", P, idx, n_tris, max_prims, split=0)
    d = _build(product.lib, "device", P, idx, n_tris, max_prims, split=0)
    assert np.array_equal(d["order"], h["order"]), "primitive order differs"
    assert np.array_equal(d["last"], h["last"]), "leaf ends differ"
    assert np.array_equal(d["info5"], h["info5"])          # interior nodes, leaves, largest leaf, depth, root reference
    assert np.array_equal(d["rb"], h["rb"])
    # the node arrays entry by entry: child references and split axes as integers, the twelve planes as numbers (a zero's sign is the one thing an atomic min does not keep)
    assert np.array_equal(d["nodes"][:, 12:15], h["nodes"][:, 12:15])
    assert np.array_equal(d["nodes"][:, :12].view(np.float32), h["nodes"][:, :12].view(np.float32))
    if n_tris <= 20000:
        orc = OracleScene()
        m = orc.add_material_matte(); orc.add_mesh(P, idx, m); orc.build_accel(0, max_prims)
        onodes = orc.bvh_nodes()
        oprims = np.zeros(n_tris, np.uint32); orc.b.lib.oracle_bvh_ordered_prims(orc.h, oprims.ctypes.data)
        ol = onodes[onodes["n_primitives"] > 0]
        assert d["leaves"] == [tuple(oprims[l["offset"]:l["offset"] + l["n_primitives"]]) for l in ol]
        if len(onodes) > 1:
            assert d["boxes"] == {tuple(n["pmin"]) + tuple(n["pmax"]) for n in onodes[1:]}
        orc.close()


def test_sah_scene_built_on_the_device_traces_like_the_host_built_one(host):
    P, idx = host.gen_random_tris(60000, 21)
    rays = np.concatenate([scenes.random_rays(80000, 5), scenes.axis_rays()])

    def scene(device_build):
        s = pbrt_hip.Scene()
        s.add_mesh(P, idx, s.add_material_matte())
        (s.build_accel_device if device_build else s.build_accel)(0, 4)
        return s
    a, b = scene(False), scene(True)
    assert scenes.hits_equal(b.intersect_batch(rays), a.intersect_batch(rays)).all()
    assert np.array_equal(a.occluded_batch(rays), b.occluded_batch(rays))
    assert np.array_equal(a.world_bound(), b.world_bound())
    sa, sb = a.accel_stats(), b.accel_stats()
    sa.pop("build_seconds"); sb.pop("build_seconds")
    assert sa == sb
    a.close(); b.close()


def test_scene_built_on_the_device_traces_like_the_host_built_one(host):
    P, idx = host.gen_random_tris(30000, 12)
    rays = np.concatenate([scenes.random_rays(80000, 4), scenes.axis_rays()])

    def scene(device_build):
        s = pbrt_hip.Scene()
        s.add_mesh(P, idx, s.add_material_matte())
        (s.build_accel_device if device_build else s.build_accel)(1, 4)
        return s
    a, b = scene(False), scene(True)
    ha, hb = a.intersect_batch(rays), b.intersect_batch(rays)
    assert scenes.hits_equal(hb, ha).all()
    assert np.array_equal(a.occluded_batch(rays), b.occluded_batch(rays))
    assert np.array_equal(a.world_bound(), b.world_bound())
    assert a.accel_stats()["interior_nodes"] == b.accel_stats()["interior_nodes"] and a.accel_stats()["leaves"] == b.accel_stats()["leaves"]
    orc = OracleScene(); orc.add_mesh(P, idx, orc.add_material_matte()); orc.build_accel(1, 4)
    want, _ = orc.intersect_batch_stats(rays)
    assert scenes.hits_equal(hb, want).all()
    with pytest.raises(pbrt_hip.PbrtHipError) as e:
        b.build_accel_device(3, 4)          # EqualCounts stays a host build
    assert e.value.code == pbrt_hip.ERR_UNSUPPORTED
    a.close(); b.close(); orc.close()


def _forest_pair(capture):
    """The same instanced scene built by the host builder and on the device: node array, leaf records, statistics."""
    out = []
    for dev in (False, True):
        s = pbrt_hip.Scene(); capture(s, dev)
        out.append((s.accel_copy(), s.accel_stats(), s))
    return out


@pytest.mark.parametrize("n_tris,instances,seed", [(1, 3, 2), (40, 7, 3), (3000, 27, 4), (20000, 200, 5)])
def test_device_forest_equals_host_forest_one_object(host, n_tris, instances, seed):
    """One object instanced K times (two-level BVH): the scene's aggregate over the TransformedPrimitives and the object's aggregate, built as one forest on the device, are
    the host builder's arrays entry by entry — and the films are the same bits."""
    spec = pbrt_hip.SceneSpec(n_tris=n_tris, seed=seed, xres=48, yres=48, spp=4)
    (h, hs, sh), (d, ds, sd) = _forest_pair(lambda s, dev: pbrt_hip.capture_spec(spec, s, host, instances=instances, device_build=dev))
    assert np.array_equal(h[0], d[0]) and np.array_equal(h[1], d[1])
    assert {k: v for k, v in hs.items() if k != "build_seconds"} == {k: v for k, v in ds.items() if k != "build_seconds"}
    fh, fd = sh.render_path(), sd.render_path()
    assert np.array_equal(fh[0].view(np.uint32), fd[0].view(np.uint32)) and (fh[2].regular_rays, fh[2].shadow_rays) == (fd[2].regular_rays, fd[2].shadow_rays)
    sh.close(); sd.close()


def test_device_forest_equals_host_forest_many_objects(host):
    """The configs[4] scene at a twentieth of its tessellation: 128 object definitions (among them meshes with alpha textures, several meshes per object), 1 100 instances and
    the courtyard's own triangles — 129 trees in one device build, against the host builder's arrays and film."""
    from pbrt_hip.sanmiguel import SanMiguelScene
    sm = SanMiguelScene(host, scale=0.05)
    (h, hs, sh), (d, ds, sd) = _forest_pair(lambda s, dev: sm.capture(s, 160, 90, 4, device_build=dev))
    assert np.array_equal(h[0], d[0]) and np.array_equal(h[1], d[1])
    assert {k: v for k, v in hs.items() if k != "build_seconds"} == {k: v for k, v in ds.items() if k != "build_seconds"}
    fh, fd = sh.render_path(), sd.render_path()
    assert np.array_equal(fh[0].view(np.uint32), fd[0].view(np.uint32)) and (fh[2].regular_rays, fh[2].shadow_rays) == (fd[2].regular_rays, fd[2].shadow_rays)
    sh.close(); sd.close()
