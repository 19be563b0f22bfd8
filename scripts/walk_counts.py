#!/usr/bin/env python3
"""Node steps, triangle tests and stack entries dropped at pop time per ray, counted by the counting build of the traversal kernel on one frame of a BASELINE configuration
at reduced spp (round 4 used it to compare the binary walk with the two-level experiment, profiles/r04_quad_experiment/).  usage: scripts/walk_counts.py [config] [spp]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-v3-rs_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pbrt_hip  # noqa: E402
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "2"])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
host = pbrt_hip.Host()
scene = pbrt_hip.Scene(device=0)
spec = pbrt_hip.SceneSpec(n_tris=cfg["n_tris"], seed=1, xres=cfg["res"], yres=cfg["res"], spp=spp, max_depth=cfg["max_depth"])
pbrt_hip.capture_spec(spec, scene, host, device_build=True)
scene.set_traversal_counting(True)
scene.render_path(max_depth=cfg["max_depth"])
c = scene.traversal_counts()
rays = c["closest"]["rays"] + c["any_hit"]["rays"]
print(json.dumps({"rays": rays,
                  "node_steps_per_ray": round((c["closest"]["nodes_passed"] + c["any_hit"]["nodes_passed"]) / rays, 2),
                  "tri_tests_per_ray": round((c["closest"]["tri_tests"] + c["any_hit"]["tri_tests"]) / rays, 2),
                  "entries_dropped_at_pop_per_ray": round(c["culled_pops"] / rays, 2)}))
