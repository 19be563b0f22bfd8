import sys, time, ctypes as C
sys.path.insert(0,'pbrt-v3-rs_amd')
import numpy as np, pbrt_hip
h=pbrt_hip.Host(); lib=pbrt_hip.default_binding().lib
lib.pbrt_hip_host_build_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
for n in (1000000, 10000000):
    P, idx = h.gen_random_tris(n, 1)
    order = np.zeros(n, np.uint32); last = np.zeros(n, np.uint32); info = np.zeros(5, np.uint64); rb = np.zeros(6, np.float32)
    for th in (1, 16):
        t=time.time(); lib.pbrt_hip_host_build_bvh(P.ctypes.data, idx.ctypes.data, n, 0, 4, th, order.ctypes.data, last.ctypes.data, None, info.ctypes.data, rb.ctypes.data); dt=time.time()-t
        print(n, 'threads', th, 'total %.2f s'%dt, flush=True)
