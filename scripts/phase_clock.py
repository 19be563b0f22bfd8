#!/usr/bin/env python3
"""Where the traversal kernel's wave cycles go, per phase of its loop (a measurement build: make -B -C pbrt-v3-rs_amd/csrc EXTRA=-DPH_PHASE_CLOCK=1 — traverse.h, wavefront.hip, phase_clock.h; the shade and texture kernels' phases are printed too).
Renders one frame of a BASELINE configuration at reduced spp with the SHIPPING kernel shapes and prints the waves' phase clocks (stderr of the library).
usage (on the GPU box, after the measurement build): scripts/phase_clock.py [config] [spp]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-v3-rs_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pbrt_hip  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "2"
cfg = dict(bench.CONFIGS[name])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
host = pbrt_hip.Host()
scene = pbrt_hip.Scene(device=0)
if name == "4":
    from pbrt_hip.sanmiguel import SanMiguelScene
    SanMiguelScene(host, scale=1.0, seed=5).capture(scene, cfg["res"], cfg["yres"], spp, device_build=True)
else:
    inst = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    spec = pbrt_hip.SceneSpec(n_tris=cfg["n_tris"] if not inst else 10000, seed=1, xres=cfg["res"], yres=cfg["res"], spp=spp, max_depth=cfg["max_depth"])
    pbrt_hip.capture_spec(spec, scene, host, device_build=True, instances=inst)
scene.render_path(max_depth=cfg["max_depth"])
scene.traversal_counts()   # flushes the phase clocks to stderr
_, _, st = scene.render_path(max_depth=cfg["max_depth"])
print(f"config {name} at {spp} spp: {st.regular_rays + st.shadow_rays} rays, traversal {1e3 * (st.extend_seconds + st.shadow_seconds):.1f} ms (measurement build: slower than shipping)")
scene.traversal_counts()
