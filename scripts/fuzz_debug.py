import sys
sys.path.insert(0,'pbrt-v3-rs_amd'); sys.path.insert(0,'tests')
import numpy as np, pbrt_hip
from oracle_binding import OracleScene, set_libm_mode
import test_fuzz_gpu as T
host = pbrt_hip.Host()
for seed in (2, 9, 17, 23, 3):
    cap, kw = T.build_case(host, seed)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    cb = cap(prod); cap(orc)
    set_libm_mode(1)
    oxyz, owt, ost, _ = orc.render_path_ex(**kw)
    set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(**kw)
    print(seed, kw, 'created', gst.light_distributions_created, ost.light_distributions_created, 'film equal', np.array_equal(gxyz.view(np.uint32), oxyz.view(np.uint32)),
          'nvox', orc.spatial_stats(), 'wb', orc.world_bound(), prod.world_bound(), 'crop', cb)
    for md in (0, 1, 2):
        k2 = dict(kw); k2['max_depth'] = md
        _, _, a = prod.render_path(**k2); _, _, b, _ = orc.render_path_ex(**k2)
        print('   max_depth', md, a.light_distributions_created, b.light_distributions_created, a.regular_rays, b.regular_rays)
