"""Test-side loader for the CPU oracle (oracle/liboracle.so).  Only tests/, smoke() and bench.py's cpu_baseline leg
may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

import pbrt_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")


class TraversalStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tri_tests", C.c_uint64)]


_binding = None


def build_oracle():
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]   # every source: a header left out would leave a stale library
    if not os.path.exists(ORACLE_LIB) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def oracle_binding():
    global _binding
    if _binding is None:
        build_oracle()
        b = pbrt_hip.Binding(ORACLE_LIB, "oracle_")
        L = b.lib
        vp = C.c_void_p
        L.oracle_intersect_batch_stats.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(TraversalStats), C.c_int]
        L.oracle_occluded_batch_stats.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(TraversalStats), C.c_int]
        L.oracle_render_path_ex.argtypes = [vp, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                            C.POINTER(C.c_float), C.POINTER(pbrt_hip.Stats), C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.oracle_record_rays.argtypes = [vp, C.c_uint64]
        L.oracle_recorded_count.argtypes = [vp, C.c_int]
        L.oracle_recorded_count.restype = C.c_uint64
        L.oracle_recorded_rays.argtypes = [vp, C.c_int, vp]
        L.oracle_bvh_node_count.argtypes = [vp]
        L.oracle_bvh_node_count.restype = C.c_uint64
        L.oracle_bvh_nodes.argtypes = [vp, vp]
        L.oracle_bvh_ordered_prims.argtypes = [vp, vp]
        L.oracle_radical_inverse.restype = C.c_float
        L.oracle_radical_inverse.argtypes = [C.c_int, C.c_uint64]
        L.oracle_scrambled_radical_inverse.restype = C.c_float
        L.oracle_scrambled_radical_inverse.argtypes = [C.c_int, C.c_uint64]
        L.oracle_sampler_value.restype = C.c_float
        L.oracle_sampler_value.argtypes = [vp, C.c_int, C.c_int, C.c_uint32, C.c_uint32]
        L.oracle_prime.restype = C.c_uint32
        L.oracle_prime_sum.restype = C.c_uint32
        L.oracle_rng_u32.argtypes = [C.c_uint64, C.c_int, vp, C.c_int]
        L.oracle_halton_perm.argtypes = [C.c_int, vp]
        L.oracle_geom_op.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_set_libm_mode.argtypes = [C.c_int]
        fp = C.POINTER(C.c_float)
        L.oracle_bsdf_probe.argtypes = [vp, C.c_uint32, C.c_int, fp, fp, fp, C.c_int, fp]
        L.oracle_spatial_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.oracle_spatial_voxel.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _binding = b
    return _binding


class OracleScene(pbrt_hip.Scene):
    """pbrt_hip.Scene driven against liboracle.so, plus the oracle-only extras."""

    def __init__(self):
        super().__init__(oracle_binding(), 0)

    def intersect_batch_stats(self, rays, threads=0):
        rays = np.ascontiguousarray(rays, dtype=pbrt_hip.RAY_DTYPE)
        hits = np.zeros(len(rays), pbrt_hip.HIT_DTYPE)
        st = TraversalStats()
        self._chk(self.b.lib.oracle_intersect_batch_stats(self.h, rays.ctypes.data, hits.ctypes.data, len(rays), C.byref(st), threads))
        return hits, st

    def occluded_batch_stats(self, rays, threads=0):
        rays = np.ascontiguousarray(rays, dtype=pbrt_hip.RAY_DTYPE)
        out = np.zeros(len(rays), np.uint8)
        st = TraversalStats()
        self._chk(self.b.lib.oracle_occluded_batch_stats(self.h, rays.ctypes.data, out.ctypes.data, len(rays), C.byref(st), threads))
        return out, st

    def render_path_ex(self, max_depth=5, rr_threshold=1.0, light_strategy=2, pixel_bounds=None, tile_size=16, tile_part=0, tile_parts=1, threads=0,
                       count_traversal=False):
        h, w = self.film_shape
        xyz = np.zeros((h, w, 3), np.float32); wt = np.zeros((h, w), np.float32)
        pb = np.ascontiguousarray(pixel_bounds if pixel_bounds is not None else self.sample_bounds, dtype=np.int32)
        st = pbrt_hip.Stats(); nvnt = (C.c_uint64 * 4)()
        self._chk(self.b.lib.oracle_render_path_ex(self.h, max_depth, C.c_float(rr_threshold), light_strategy, pb.ctypes.data_as(C.POINTER(C.c_int)), tile_size,
                                                  tile_part, tile_parts, xyz.ctypes.data_as(C.POINTER(C.c_float)), wt.ctypes.data_as(C.POINTER(C.c_float)),
                                                  C.byref(st), threads, 1 if count_traversal else 0, nvnt))
        return xyz, wt, st, list(nvnt)

    def bsdf_probe(self, material, op, wo=(0, 0, 1), wi=(0, 0, 1), u=(0.5, 0.5), flags=31):
        """BSDF of `material` in the canonical frame: op 0 -> (f rgb, pdf), op 1 -> sample_f (f rgb, pdf, wi xyz, type), op 2 -> counts."""
        out = np.zeros(8, np.float32)
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float))
        self._chk(self.b.lib.oracle_bsdf_probe(self.h, material, op, f(wo), f(wi), f(u), flags, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def spatial_stats(self):
        """(voxel resolution xyz, distributions created) of the last spatial render."""
        out = (C.c_uint64 * 4)()
        self._chk(self.b.lib.oracle_spatial_stats(self.h, out))
        return tuple(out[:3]), int(out[3])

    def spatial_voxel(self, pi, n_lights):
        """compute_distribution for voxel pi: (func[n], cdf[n+1], func_int)."""
        func = np.zeros(n_lights, np.float32); cdf = np.zeros(n_lights + 1, np.float32); fi = C.c_float()
        pi = np.ascontiguousarray(pi, dtype=np.int32)
        self._chk(self.b.lib.oracle_spatial_voxel(self.h, pi.ctypes.data_as(C.POINTER(C.c_int)), func.ctypes.data_as(C.POINTER(C.c_float)),
                                                 cdf.ctypes.data_as(C.POINTER(C.c_float)), C.byref(fi)))
        return func, cdf, fi.value

    def record_rays(self, cap):
        self.b.lib.oracle_record_rays(self.h, cap)

    def recorded_rays(self, shadow=False):
        n = self.b.lib.oracle_recorded_count(self.h, 1 if shadow else 0)
        rays = np.zeros(n, pbrt_hip.RAY_DTYPE)
        if n:
            self.b.lib.oracle_recorded_rays(self.h, 1 if shadow else 0, rays.ctypes.data)
        return rays

    def bvh_nodes(self):
        n = self.b.lib.oracle_bvh_node_count(self.h)
        dt = np.dtype([("pmin", "<f4", 3), ("pmax", "<f4", 3), ("offset", "<u4"), ("n_primitives", "<u2"), ("axis", "u1"), ("pad", "u1")])
        nodes = np.zeros(n, dt)
        self.b.lib.oracle_bvh_nodes(self.h, nodes.ctypes.data)
        return nodes


def set_libm_mode(mode):
    oracle_binding().lib.oracle_set_libm_mode(mode)
