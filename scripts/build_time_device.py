"""Build + transfer time of the device builders (SAH, HLBVH) against the host builders: python scripts/build_time_device.py [n_tris ...] (needs a GPU)."""
import ctypes as C, os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "pbrt-v3-rs_amd"))
import numpy as np, pbrt_hip
host = pbrt_hip.Host(); lib = pbrt_hip.default_binding().lib
lib.pbrt_hip_host_build_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.pbrt_hip_device_build_bvh.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
for n in [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000]:
    P, idx = host.gen_random_tris(n, 1)
    order = np.zeros(n, np.uint32); last = np.zeros(n, np.uint32); info = np.zeros(5, np.uint64); rb = np.zeros(6, np.float32)
    out = {}; host_order = {}
    for name, sm in (("host SAH", 0), ("host HLBVH", 1)):
        t = time.time(); rc = lib.pbrt_hip_host_build_bvh(P.ctypes.data, idx.ctypes.data, n, sm, 4, 0, order.ctypes.data, last.ctypes.data, None, info.ctypes.data, rb.ctypes.data); out[name] = time.time() - t
        assert rc == 0
        host_order[sm] = order.copy()
    secs = C.c_double(0); same = {}; inner = {}
    for name, sm in (("device HLBVH", 1), ("device SAH", 0)):
        for rep in range(2):   # the first call also pays the device's start-up
            t = time.time(); rc = lib.pbrt_hip_device_build_bvh(0, P.ctypes.data, idx.ctypes.data, n, sm, 4, order.ctypes.data, last.ctypes.data, None, info.ctypes.data, rb.ctypes.data, C.byref(secs)); out[name] = time.time() - t
            assert rc == 0
        same[name] = bool(np.array_equal(order, host_order[sm])); inner[name] = round(secs.value, 3)
    print(n, "triangles:", {k: round(v, 3) for k, v in out.items()}, "device-internal", inner, "s; same order as the host build:", same, "depth", int(info[3]), flush=True)
