"""Diagnostic: projection / goniometric lights, one configuration at a time (python scripts/light_probe.py; needs a GPU)."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "pbrt-v3-rs_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import pbrt_hip, scenes
host = pbrt_hip.Host()
I4 = (np.eye(4, dtype=np.float32).reshape(16),) * 2
base = scenes.cornell_like(host, sigma=10.0)
rng = np.random.default_rng(5)
slide = rng.uniform(0.0, 1.0, (12, 20, 3)).astype(np.float32)
a = host.compose(host.compose(I4, host.translate([0.4, -0.5, 0.9])), host.rotate(170.0, [1, 0.1, 0]))
for name, add in (("proj_map", lambda s: s.add_light_projection((6, 5, 4), a[0], a[1], 55.0, slide)), ("proj_none", lambda s: s.add_light_projection((6, 5, 4), a[0], a[1], 55.0, None)),
                  ("gonio_map", lambda s: s.add_light_goniometric((6, 5, 4), a[0], a[1], slide)), ("gonio_none", lambda s: s.add_light_goniometric((6, 5, 4), a[0], a[1], None))):
    for strategy in (0, 1, 2):
        print(name, strategy, "...", flush=True)
        s = pbrt_hip.Scene(); add(s); base(s)
        xyz, wt, st = s.render_path(max_depth=4, light_strategy=strategy)
        print("   ok mean", float(xyz.mean()), flush=True)
