"""What one rank of an N-GPU run has to do, measured on ONE GPU: configs[3] (or --config) rendered as tile part 0 of N for N = 1, 2, 4, 8.  The tiles are dealt round-robin, so every
part is the same amount of work; t(1) / (N t(N)) is the strong-scaling efficiency an N-GPU run can reach before the film-tile gather is added.  Not a scaling measurement."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pbrt-v3-rs_amd"))
import pbrt_hip  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="3")
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
host = pbrt_hip.Host()
spec = pbrt_hip.SceneSpec(n_tris=cfg["n_tris"], seed=1, xres=cfg["res"], yres=cfg["res"], spp=cfg["spp"], max_depth=cfg["max_depth"])
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")   # torch's context first, as bench.py does
with pbrt_hip.Scene() as s:
    pbrt_hip.capture_spec(spec, s, host)
    out = {}
    for n in (1, 2, 4, 8):
        floats = s.tile_buffer_floats(16, 0, n)
        buf = torch.empty(floats, dtype=torch.float32, device="cuda")
        s.render_path_tiles_device(buf.data_ptr(), max_depth=cfg["max_depth"], tile_part=0, tile_parts=n)   # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rays = 0
        for _ in range(a.steps):
            st = s.render_path_tiles_device(buf.data_ptr(), max_depth=cfg["max_depth"], tile_part=0, tile_parts=n)
            rays = st.regular_rays + st.shadow_rays
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        out[n] = {"ms": round(dt * 1e3, 2), "rays": int(rays), "Mrays_per_s_of_this_rank": round(rays / dt / 1e6, 1)}
        del buf
    t1 = out[1]["ms"]
    for n in out:
        out[n]["efficiency_before_gather"] = round(t1 / (n * out[n]["ms"]), 3)
    print(json.dumps({"config": a.config, "parts": out}))
