"""not gpu: the N>1 protocol of bench.py (tile t on rank t % N, per-rank film contributions gathered on rank 0, merged in
tile order) exercised with world_size 2 over gloo.  No GPU here, so the CPU oracle stands in for the per-rank renderer:
what is under test is the partition + gather + merge logic, which is identical for the device path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.join(ROOT, "pbrt-v3-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pbrt_hip
    from oracle_binding import OracleScene
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    host = pbrt_hip.Host()
    spec = pbrt_hip.SceneSpec(n_tris=800, seed=9, xres=70, yres=45, spp=4)
    s = OracleScene()
    pbrt_hip.capture_spec(spec, s, host)
    xyz, wt, st, _ = s.render_path_ex(tile_part=rank, tile_parts=world, threads=2)
    mine = torch.from_numpy(np.concatenate([xyz.ravel(), wt.ravel()]))
    gathered = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)
    rays = torch.tensor([st.regular_rays + st.shadow_rays], dtype=torch.int64)
    dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)   # bench.py takes the MAX of the per-rank times
    if rank == 0:
        acc = torch.zeros_like(mine)
        for g in gathered:           # merge in increasing rank = increasing tile index within an overlap (box filter: exact)
            acc += g
        full_xyz, full_wt, full_st, _ = s.render_path_ex(threads=2)
        np.save(out_path, np.array([
            np.array_equal(acc.numpy()[: xyz.size].view(np.uint32), full_xyz.ravel().view(np.uint32)),
            np.array_equal(acc.numpy()[xyz.size:], full_wt.ravel()),
            int(rays.item()) == full_st.regular_rays + full_st.shadow_rays,
            t.item() == float(world),
        ]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharding_equals_single_rank(tmp_path):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ok = np.load(out)
    assert ok.all(), ok


def test_bench_without_a_launcher_starts_its_ranks_and_relays_their_return_code():
    """`python bench.py --gpus 2` (no WORLD_SIZE): bench.py itself starts torch.distributed.run with two ranks as a child process and hands back what they
    return.  There is no GPU here, so both ranks stop with "needs a GPU" — which proves that the ranks were started (the old behaviour was to refuse at once)
    and that a failing rank reaches the caller as a non-zero exit code with nothing on stdout."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: tests/test_multi_rank_gpu.py covers the launcher")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr and "torch.distributed" in r.stderr, r.stderr[-3000:]   # said by a rank that the elastic launcher started
    assert r.stdout.strip() == ""
