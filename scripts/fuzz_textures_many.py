"""Wider sweep of tests/test_fuzz_textures_gpu.py's generator: python scripts/fuzz_textures_many.py FIRST LAST (needs a GPU)."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "pbrt-v3-rs_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import pbrt_hip
import test_fuzz_textures_gpu as T
host = pbrt_hip.Host()
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []; skipped = 0
for seed in range(first, last):
    if (seed - first) % 50 == 0: print('progress: seed', seed, 'bad so far', bad, flush=True)
    ok, info = T.run_case(host, seed)
    if ok is None: skipped += 1
    elif not ok: bad.append(seed); print("MISMATCH seed", seed, info, flush=True)
    if seed % 100 == 0: print("...", seed, flush=True)
print("checked", last - first - skipped, "scenes (", skipped, "refused ); mismatches:", bad)
