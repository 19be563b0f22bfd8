"""One-off wider sweep of tests/test_fuzz_gpu.py's generator: python scripts/fuzz_many.py FIRST LAST (needs a GPU)."""
import sys
sys.path.insert(0, 'pbrt-v3-rs_amd'); sys.path.insert(0, 'tests')
import numpy as np, pbrt_hip
from oracle_binding import OracleScene, set_libm_mode
import test_fuzz_gpu as T
host = pbrt_hip.Host()
first, last = int(sys.argv[1]), int(sys.argv[2])
big = len(sys.argv) > 3 and sys.argv[3] == "big"
bad = []
for seed in range(first, last):
    if (seed - first) % 50 == 0: print('progress: seed', seed, 'bad so far', bad, flush=True)
    cap, kw = T.build_case(host, seed, big)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    try:
        cb = cap(prod); cap(orc)
    except pbrt_hip.PbrtHipError as e:
        print(seed, 'setup error', e); prod.close(); orc.close(); continue
    if (cb[2] - cb[0]) * (cb[3] - cb[1]) <= 0:
        prod.close(); orc.close(); continue
    set_libm_mode(1)
    oxyz, owt, ost, _ = orc.render_path_ex(**kw)
    set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(**kw)
    ok = (gst.regular_rays, gst.shadow_rays, gst.paths_total, gst.paths_zero_radiance, gst.light_distributions_created) == \
         (ost.regular_rays, ost.shadow_rays, ost.paths_total, ost.paths_zero_radiance, ost.light_distributions_created)
    ok = ok and np.array_equal(gwt.view(np.uint32), owt.view(np.uint32)) and np.array_equal(gxyz.view(np.uint32), oxyz.view(np.uint32))
    if not ok:
        nb = int((gxyz.view(np.uint32) != oxyz.view(np.uint32)).any(axis=2).sum())
        bad.append(seed); print('MISMATCH seed', seed, kw, 'pixels', nb, gst.as_dict(), ost.as_dict(), flush=True)
    prod.close(); orc.close()
print('checked', last - first, 'bad', bad)
