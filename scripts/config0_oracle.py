"""BASELINE.json configs[0] (scenes/shapes/sphere.pbrt, PathIntegrator maxdepth 4, 16 spp, 800 x 400) in the CPU oracle — "CPU reference only (plumbing)":
the product renders triangles, the Sphere exists in the oracle alone.  Prints one JSON line; optional argument: a .pfm path for the image."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-v3-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pbrt_hip  # noqa: E402
from oracle_binding import oracle_binding  # noqa: E402
from test_oracle_sphere import configs0_scene  # noqa: E402

host = pbrt_hip.Host()
with pbrt_hip.Scene(oracle_binding()) as s:
    configs0_scene(s, host, 800, 400, 16)
    t0 = time.perf_counter()
    xyz, wt, st = s.render_path(max_depth=4)
    dt = time.perf_counter() - t0
    rgb = s.film_to_rgb(xyz, wt)
d = st.as_dict()
rays = d["regular_rays"] + d["shadow_rays"]
print(json.dumps({"workload": "configs[0]: scenes/shapes/sphere.pbrt, path maxdepth 4, halton 16 spp, 800x400, constant Kd for the uv-grid map", "engine": "CPU oracle",
                  "threads": os.cpu_count(), "seconds": round(dt, 3), "rays": rays, "Mrays_per_s": round(rays / dt / 1e6, 2), "mean_rgb": [round(float(v), 4) for v in rgb.mean((0, 1))]}))
if len(sys.argv) > 1:
    with open(sys.argv[1], "wb") as f:
        f.write(b"PF\n800 400\n-1.0\n"); f.write(np.ascontiguousarray(rgb[::-1]).astype("<f4").tobytes())
