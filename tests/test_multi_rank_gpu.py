"""-m gpu: the N>1 device path (per-rank tile buffers -> gather -> merge_tiles_device) rehearsed with 2 ranks sharing the one
GPU of the test box (gloo carries the tile buffers through host memory; on a multi-GPU node bench.py uses RCCL instead).
The merged frame must be bit-identical to the 1-rank frame."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    return json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])


def test_two_ranks_one_gpu_film_identical():
    common = ["--config", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline-count", "--spp", "4", "--res", "200", "--n-tris", "20000"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"] + common, capture_output=True, text=True, env=env, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = _last_json(one.stdout), _last_json(two.stdout)
    assert b["n_gpus"] == 2 and a["n_gpus"] == 1
    assert b["scaling"] == "strong"   # the N > 1 default: the N = 1 frame split over the ranks
    assert a["film_sha256"] == b["film_sha256"]
    assert a["config"]["rays_per_frame"] == b["config"]["rays_per_frame"]
    # the short weak-scaling leg reported alongside renders spp x N
    assert b["weak_alongside"]["spp"] == 8 and b["weak_alongside"]["value"] > 0


def test_two_ranks_weak_scaling_doubles_the_samples():
    """--scaling weak: spp x N, so per-rank work equals the one-rank frame; the merged 2-rank frame equals a one-rank render of
    the same 2x-spp frame."""
    common = ["--config", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline-count", "--res", "128", "--n-tris", "5000"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spp", "8"] + common, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29534",
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--scaling", "weak", "--spp", "4"] + common, capture_output=True, text=True, env=env, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = _last_json(one.stdout), _last_json(two.stdout)
    assert b["scaling"] == "weak" and b["n_gpus"] == 2
    assert a["film_sha256"] == b["film_sha256"] and a["config"]["rays_per_frame"] == b["config"]["rays_per_frame"]


def test_two_ranks_at_config3_size():
    """The N > 1 default workload — configs[3]: 10 M triangles, 2048 x 2048 @ 64 spp, strong scaling — rehearsed with 2 ranks on the one GPU:
    the merged film and the ray counts equal the one-rank frame's."""
    common = ["--config", "3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline-count"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, env=env, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29535",
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-weak-leg"] + common, capture_output=True, text=True, env=env, timeout=1200)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = _last_json(one.stdout), _last_json(two.stdout)
    assert a["config"]["baseline_config"] == b["config"]["baseline_config"] == "3" and b["scaling"] == "strong" and b["n_gpus"] == 2
    assert a["film_sha256"] == b["film_sha256"]
    assert a["config"]["rays_per_frame"] == b["config"]["rays_per_frame"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (what the driver's command line looks like): bench.py starts torch.distributed.run as a child before
    it touches torch or HIP and relays the ranks' single JSON line; the film equals the one-rank film."""
    common = ["--config", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline-count", "--spp", "4", "--res", "160", "--n-tris", "8000"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-weak-leg"] + common, capture_output=True, text=True, env=env, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [l for l in two.stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1, lines   # ONE JSON line on stdout
    a, b = _last_json(one.stdout), json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "strong"
    assert a["film_sha256"] == b["film_sha256"] and a["config"]["rays_per_frame"] == b["config"]["rays_per_frame"]
    # what a first real N-GPU line needs to be diagnosable (round-3 review): where the step's time went, every rank's share, what the collective library saw
    assert b["rccl"]["world_size"] == 2 and b["rccl"]["backend"] == "gloo"
    rk = b["ranks"]
    assert len(rk["render_ms"]["per_rank"]) == 2 and rk["render_ms"]["min"] <= rk["render_ms"]["mean"] <= rk["render_ms"]["max"]
    assert 1.0 <= rk["imbalance"] < 2.0 and sum(rk["rays_per_rank"]) == b["config"]["rays_per_frame"]
    assert b["stage_ms"]["render"] > 0.0 and b["stage_ms"]["gather_merge"] > 0.0
    assert b["stage_ms"]["render"] + b["stage_ms"]["gather_merge"] <= b["ms_per_step"] * 1.05
    assert "ranks" not in a and a["stage_ms"]["gather_merge"] >= 0.0


def test_bench_multi_handle_two_contexts():
    """--multi-handle: ONE process, one pbrt_hip_scene_create_multi handle over 2 contexts (sharing GPU 0 under --backend gloo), tiles gathered and merged
    inside the library; same film and ray count as the one-device frame."""
    common = ["--config", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline-count", "--spp", "4", "--res", "160", "--n-tris", "8000"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--multi-handle", "--backend", "gloo", "--no-weak-leg"] + common,
                         capture_output=True, text=True, env=env, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = _last_json(one.stdout), _last_json(two.stdout)
    assert b["n_gpus"] == 2 and "multi-device handle" in b["config"]["tiles"]
    assert a["film_sha256"] == b["film_sha256"] and a["config"]["rays_per_frame"] == b["config"]["rays_per_frame"]


def test_split_traversal_env_gives_the_same_film():
    """PBRT_HIP_SPLIT_TRAVERSAL (measurement aid: one launch per ray kind) must trace every shadow ray: film and ray counts equal the default path's.
    (The any-hit launch used to find the round's queue heads already drained by the closest-hit launch.)"""
    common = ["--config", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline-count", "--spp", "4", "--res", "160", "--n-tris", "8000"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PBRT_HIP_SPLIT_TRAVERSAL", None)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, env=dict(env, PBRT_HIP_SPLIT_TRAVERSAL="1"), timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = _last_json(one.stdout), _last_json(two.stdout)
    assert a["film_sha256"] == b["film_sha256"] and a["config"]["rays_per_frame"] == b["config"]["rays_per_frame"]


def test_ab_slots_and_binning_modes_give_the_same_film():
    """The kernels kept as A/B slots (PBRT_HIP_TRAV_VARIANT=1: the round-3 loop shape; PBRT_HIP_INST_VARIANT=1: the 4-wave instancing kernel), every ray-binning mode
    (PBRT_HIP_SORT_RAYS 0 / 2 / 3), the walk without the shade-side work queues (PBRT_HIP_MATERIAL_QUEUES=0) and the alpha-mask thresholds (PBRT_HIP_ALPHA_MIN 0 / 20) trace the
    frame the shipping configuration traces: the order in which a round's rays are traced, the loop thresholds and the occupancy are free
    choices, the film is not."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("PBRT_HIP_TRAV_VARIANT", "PBRT_HIP_INST_VARIANT", "PBRT_HIP_SORT_RAYS", "PBRT_HIP_MATERIAL_QUEUES", "PBRT_HIP_ALPHA_MIN"):
        env.pop(k, None)

    def run(extra_args, extra_env):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline-count", "--spp", "4", "--res", "160"] + extra_args,
                           capture_output=True, text=True, env=dict(env, **extra_env), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        d = _last_json(r.stdout)
        return d["film_sha256"], d["config"]["rays_per_frame"]

    flat = ["--n-tris", "8000"]
    base = run(flat, {})
    for e in ({"PBRT_HIP_TRAV_VARIANT": "1"}, {"PBRT_HIP_SORT_RAYS": "0"}, {"PBRT_HIP_SORT_RAYS": "2"}, {"PBRT_HIP_SORT_RAYS": "3"}):
        assert run(flat, e) == base, e
    inst = ["--n-tris", "500", "--instances", "40"]
    base_i = run(inst, {})
    assert run(inst, {"PBRT_HIP_INST_VARIANT": "1"}) == base_i
    # round 4: the shade-side work queues (who shades which path when) and the company an alpha-mask verdict waits for are free choices too
    mixed = ["--n-tris", "8000", "--material", "mixed"]
    base_m = run(mixed, {})
    assert run(mixed, {"PBRT_HIP_MATERIAL_QUEUES": "0"}) == base_m
    textured = ["--n-tris", "8000", "--material", "textured"]
    assert run(textured, {"PBRT_HIP_MATERIAL_QUEUES": "0"}) == run(textured, {})
    c4 = ["--config", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline-count", "--spp", "2", "--sm-scale", "0.05"]

    def run4(extra_env):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + c4, capture_output=True, text=True, env=dict(env, **extra_env), timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        d = _last_json(r.stdout)
        return d["film_sha256"], d["config"]["rays_per_frame"]
    base_4 = run4({})   # the San-Miguel-shaped scene at a twentieth of its tessellation: instances + alpha masks + every material class
    for e in ({"PBRT_HIP_ALPHA_MIN": "0"}, {"PBRT_HIP_ALPHA_MIN": "20"}, {"PBRT_HIP_MATERIAL_QUEUES": "0"}):
        assert run4(e) == base_4, e


def test_bench_nccl_code_path_with_one_rank():
    """The N > 1 code path of bench.py over the REAL backend — `init_process_group("nccl")` (= RCCL), barrier, `dist.gather` of the device tile buffer, all_reduce of the timings, merge of
    the gathered buffers — exercised with the one rank a one-GPU box has (`--force-dist` under torch.distributed.run --nproc-per-node 1): the very calls an 8-GPU node makes, and the
    same film as the plain run."""
    common = ["--config", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline-count", "--spp", "4", "--res", "160", "--n-tris", "8000"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    rc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29541",
                         os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl"] + common, capture_output=True, text=True, env=env, timeout=900)
    assert rc.returncode == 0, rc.stderr[-3000:]
    a, b = _last_json(one.stdout), _last_json(rc.stdout)
    assert a["film_sha256"] == b["film_sha256"] and a["config"]["rays_per_frame"] == b["config"]["rays_per_frame"]
