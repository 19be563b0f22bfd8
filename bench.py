#!/usr/bin/env python3
"""bench.py — Mrays/s of the MI355X wavefront path tracer on BASELINE.json's configurations (synthetic random-triangle scenes).

  python bench.py --gpus N --steps K --warmup W
  N > 1 runs one rank per GPU.  Started under torch.distributed.run (WORLD_SIZE set) it is one of the ranks; started plainly it launches
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>` as a CHILD process BEFORE torch or
  HIP are touched (a process that has initialised the GPU is never re-executed), relays the ranks' one JSON line and exits with their return code.
  --multi-handle: ONE process, one handle over the N GPUs (pbrt_hip_scene_create_multi: one host thread per GPU, film tiles gathered inside the
  library with ncclSend / ncclRecv); same partition, same film.

Workload (`--config`, named in the JSON line's config.workload):
  2 (default at N = 1)  configs[2]: 4.3 M triangles, PathIntegrator maxdepth 8, 1024x1024 @ 256 spp — the largest single-GPU configuration.  The
                        Ganesha PLY is not available offline; the stand-in is the synthetic generator at the same triangle count.
  3 (default at N > 1)  configs[3]: 10 M triangles, 2048x2048 @ 64 spp, tiles sharded over the ranks, film tiles gathered with RCCL.  STRONG scaling:
                        the frame is the same for every N; a short weak-scaling leg (spp x N) is reported alongside under `weak_alongside`.
  1                     configs[1]: 100 k triangles, 512x512 @ 64 spp.          1M: the north-star 1 M-triangle scene at 512x512 @ 64 spp.
  4                     configs[4]: "San Miguel (instanced, ~10 M tris, many materials), 1920x1080 @ 512 spp" on the generator of pbrt_hip/sanmiguel.py (the asset
                        is not available offline): two-level BVH, TransformedPrimitive traversal, alpha masks, textured general materials, spatial light sampling.
  --n-tris/--res/--spp/--max-depth override single values (the workload is then labelled "custom").

A step = one frame: SamplerIntegrator::render of the whole image (raygen -> [traverse, shade] x (maxdepth+1) -> film), film read back to the host.
Scene, BVH and sampler tables are resident in HBM before the timed region.  With N ranks the frame's 16x16 tiles are dealt round-robin
(tile t -> rank t % N, the reference's tile enumeration), every rank renders its tiles, the per-tile film buffers are gathered on rank 0 over RCCL
(the path's one exchange step) and merged there in tile order.

Prints ONE JSON line (rank 0).
`roofline` describes the dominant kernel (ph::traverse_kernel: BVH traversal + triangle tests; one launch per wavefront round serves that round's
closest-hit and any-hit rays):
  achieved  = ALGORITHMIC bytes / kernel time: 32 B per reference-format node visit + 48 B per triangle test + 64 B ray in / hit out (33 B for any-hit
              rays), SURVEY §8d, counted by an untimed counting build of the same kernel on the same frame; time = HIP events around every traversal
              launch of the timed steps, recorded on the library's own stream.
  traffic   = memory-side bytes per launch (rocprofv3 PMC FETCH_SIZE + WRITE_SIZE, i.e. what left the L2s towards Infinity Cache / HBM), read from
              profiles/r02_traffic.json when that file holds a PMC run of this exact workload (scripts/pmc_profile.sh writes it), else null + the reason.
  frac      = traffic / avg launch time / peak: the fraction of the HBM peak the kernel draws at the memory side.  The tree's upper levels are re-read
              by every ray and are served by L2 / Infinity Cache, so the algorithmic rate (`frac_algorithmic` = achieved / peak) is not an HBM
              fraction and can exceed 1; it is reported next to it.  Without a matching PMC run of the workload frac is null.
`cpu_baseline` times the CPU oracle (oracle/, a port of the reference algorithm — the Rust reference cannot be built here) on a bounded sample of the
same workload (same scene and resolution, the first few Halton samples of every pixel, about 150 M rays) on this box's host cores.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pbrt-v3-rs_amd"))

L1_LOOKUP_PEAK_G = 880.0   # G L1 (TCP) tag look-ups per second the chip sustains for one random 64-B record per lane read as 4 x dwordx4 (scripts/calib/node_fetch.hip, mode A, L2-resident table)
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
INFINITY_CACHE_BYTES = 256 << 20
L2_PEAK_GBS = 34500.0      # aggregate L2 read bandwidth of the 8 XCDs (MI355X_MICROARCH.md)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_traffic.json")

CONFIGS = {  # BASELINE.json `configs`, on the synthetic generator of SURVEY §8d
    "1": dict(n_tris=100_000, res=512, spp=64, max_depth=5, label="configs[1]: synthetic 100 k random triangles, single BVH, 512x512 @ 64 spp"),
    "2": dict(n_tris=4_300_000, res=1024, spp=256, max_depth=8,
              label="configs[2]: 4.3 M triangles (synthetic stand-in of the same size for the Ganesha PLY, which is not available offline), PathIntegrator depth 8, 1024x1024 @ 256 spp"),
    "3": dict(n_tris=10_000_000, res=2048, spp=64, max_depth=5, label="configs[3]: synthetic 10 M triangles, 2048x2048 @ 64 spp, tiles sharded over the ranks, RCCL film-tile gather"),
    "1M": dict(n_tris=1_000_000, res=512, spp=64, max_depth=5, label="north-star 1 M-triangle scene: synthetic 1 M random triangles, 512x512 @ 64 spp"),
    # configs[4]: the San Miguel asset is not available offline; pbrt_hip/sanmiguel.py generates a scene of the same shape (n_tris is filled in from the generator)
    "4": dict(n_tris=0, res=1920, yres=1080, spp=512, max_depth=5,
              label="configs[4]: San-Miguel-shaped synthetic scene (the asset is not available offline): 128 objects, 1 100 instances = 10.2 M instanced + 0.4 M top-level triangles, "
                    "26 materials (image maps, procedural textures, bump maps, alpha-masked foliage, glass / metal / uber / translucent / mix), radiance-map sky + sun + point + 8 emissive "
                    "triangles (spatial light distribution), 1920x1080 @ 512 spp"),
}


def host_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota and by the box's per-GPU CPU share (16)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("PBRT_HIP_CPU_THREADS", "16"))))


def workload_key(n_tris, res, spp, max_depth, seed, yres=None, instances=0):
    return [int(n_tris), int(res), int(spp), int(max_depth), int(seed)] + ([int(yres)] if yres and yres != res else []) + (["instances", int(instances)] if instances else [])


def library_identity():
    """(sha256 over the library's sources — every file of pbrt-v3-rs_amd/csrc, which is what decides the kernels; a rebuild of the same sources keeps it —, the PBRT_HIP_*
    tuning variables in force): what a PMC record must have been measured on to describe this run."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "pbrt-v3-rs_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".h", ".hip", ".cpp", ".inc")) or name == "Makefile":
            with open(os.path.join(src, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest(), {k: v for k, v in sorted(os.environ.items()) if k.startswith("PBRT_HIP_") and k not in ("PBRT_HIP_CPU_THREADS", "PBRT_HIP_LIB")}


def find_traffic(key):
    """The PMC record of this workload from profiles/r04_traffic.json — measured on THIS library build under this tuning environment —, or (None, reason)."""
    if not os.path.exists(TRAFFIC_FILE):
        return None, f"{os.path.relpath(TRAFFIC_FILE, ROOT)} does not exist"
    with open(TRAFFIC_FILE) as f:
        tj = json.load(f)
    for e in tj.get("entries", []):
        if e.get("workload") == key:
            if "traversal" not in e:
                return None, "the PMC record of this workload has no traversal kernels"
            sha, env = library_identity()
            if e.get("lib_sha256") != sha:
                return None, f"the PMC record of this workload was measured on another build of the library (record {str(e.get('lib_sha256'))[:12]}, running {sha[:12]}): re-run scripts/pmc_profile.sh"
            if e.get("env", {}) != env:
                return None, f"the PMC record of this workload was measured under another tuning environment ({e.get('env')} vs {env})"
            return e, None
    return None, f"no PMC run of workload {key} in {os.path.relpath(TRAFFIC_FILE, ROOT)} (have: {[e.get('workload') for e in tj.get('entries', [])]})"


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a fresh child (this process has not imported torch nor touched HIP), relay
    their output and return code."""
    import socket
    import subprocess
    # under rocprofv3 the profiler's preloaded library has already initialised the GPU in THIS process: starting a launcher from it is the exec-after-GPU-init
    # this pool forbids.  Profile multi-rank runs by putting rocprofv3 around each rank (before any GPU call), or use --multi-handle (one process).
    if any(k.startswith("ROCPROFILER_") or k.startswith("ROCPROF_") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        sys.stderr.write("bench.py --gpus N > 1 cannot start its own ranks under rocprofv3 (the GPU is already initialised in this process); "
                         "profile each rank separately or use --multi-handle\n")
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout:   # rank 0 prints the one JSON line; anything else a rank writes to stdout goes to stderr so that stdout stays one line
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return p.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="BASELINE.json configuration; default 2 at N = 1, 3 at N > 1")
    ap.add_argument("--n-tris", type=int, default=None)
    ap.add_argument("--res", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--max-depth", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ply", default=None, help="a PLY mesh to render instead of the synthetic triangles (e.g. the Ganesha mesh of configs[2], which is not available offline): "
                                               "normalised into the synthetic scene's [-1, 1]^3 so that camera and light are unchanged")
    ap.add_argument("--cpu-spp", type=int, default=0, help="spp of the bounded CPU-baseline sample (same scene, same resolution); 0 = about 150 M rays' worth")
    ap.add_argument("--host-build", action="store_true", help="build the BVH with the library's host builder instead of on the GPU (the same tree either way; outside the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-count", action="store_true")
    ap.add_argument("--material", default="matte", choices=["matte", "plastic", "glass", "metal", "uber", "mixed", "textured"],
                    help="material of every triangle; anything but matte runs the general-BSDF shade kernel (not the headline workload)")
    ap.add_argument("--instances", type=int, default=0,
                    help="K > 0: the triangles become one object instanced K times (two-level BVH, TransformedPrimitive path); not the headline workload")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N>1: strong = the N=1 frame split over N ranks (headline), weak = spp x N (per-GPU work fixed)")
    ap.add_argument("--no-weak-leg", action="store_true", help="N>1, strong: skip the short weak-scaling leg reported alongside")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = functional rehearsal of the N>1 path on ONE GPU: all ranks share device 0 and film tiles travel through host memory")
    ap.add_argument("--sm-scale", type=float, default=1.0, help="--config 4: tessellation scale of the generated scene (triangle counts ~ scale; 1 = the configuration's size)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the N > 1 code path (process group, barrier, film-tile gather over RCCL, merge of the gathered buffers) with the ranks there are, "
                         "also with ONE — under `python -m torch.distributed.run --nproc-per-node 1`: the same torch.distributed / RCCL calls as on an 8-GPU node")
    ap.add_argument("--multi-handle", action="store_true",
                    help="N>1 from ONE process: one pbrt_hip_scene_create_multi handle over the N GPUs, the film-tile gather inside the library (RCCL send / recv); "
                         "with --backend gloo the N contexts share GPU 0 (rehearsal)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.multi_handle:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.multi_handle and world > 1:
        raise SystemExit("bench.py --multi-handle is one process over N GPUs: do not start it under torch.distributed.run")
    if not args.multi_handle and world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} under a launcher with WORLD_SIZE={world}: the two must agree")

    n_gpus = args.gpus if args.multi_handle else world
    cfg_name = args.config or ("2" if n_gpus == 1 else "3")
    cfg = dict(CONFIGS[cfg_name])
    custom = []
    for k, v in (("n_tris", args.n_tris), ("res", args.res), ("spp", args.spp), ("max_depth", args.max_depth)):
        if v is not None and v != cfg[k]:
            cfg[k] = v
            custom.append(k)
    n_tris, res, spp, max_depth = cfg["n_tris"], cfg["res"], cfg["spp"], cfg["max_depth"]
    yres = cfg.get("yres", res) if args.res is None else res

    import numpy as np
    import torch
    import pbrt_hip

    dist = None
    if args.backend == "gloo":
        local_rank = 0  # rehearsal mode: every rank drives GPU 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    use_dist = world > 1 or (args.force_dist and "RANK" in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    host = pbrt_hip.Host()
    tile_size = 16
    geometry = None
    if args.ply:
        P, idx = pbrt_hip.read_ply(args.ply)
        lo, hi = P.min(axis=0), P.max(axis=0)
        P = ((P - (lo + hi) * 0.5) * np.float32(2.0 / float((hi - lo).max()))).astype(np.float32)
        geometry = (np.ascontiguousarray(P), idx)
        n_tris = len(idx) // 3
        custom.append(f"mesh from {os.path.basename(args.ply)}")

    sm = None
    if cfg_name == "4":
        if args.ply or args.instances or args.material != "matte":
            raise SystemExit("--config 4 is a fixed scene: --ply / --instances / --material do not apply")
        from pbrt_hip.sanmiguel import SanMiguelScene
        sm = SanMiguelScene(host, scale=args.sm_scale, seed=args.seed + 4)
        n_tris = sm.counts()["total_triangles_as_instanced"]
        if args.sm_scale != 1.0:
            custom.append(f"tessellation scale {args.sm_scale}")

    def capture(scene, frame_spp, device_build):
        """The workload's scene into `scene` (the product's handle, or the oracle's for the CPU baseline); returns nothing, builds the accelerator."""
        nonlocal geometry
        if sm is not None:
            sm.capture(scene, res, yres, frame_spp, device_build=device_build)
            return
        spec = pbrt_hip.SceneSpec(n_tris=n_tris, seed=args.seed, xres=res, yres=yres, spp=frame_spp, max_depth=max_depth, material=args.material)
        geometry = pbrt_hip.capture_spec(spec, scene, host, geometry=geometry, instances=args.instances, device_build=device_build)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_scalars(vals, op, dtype):
        t = torch.tensor(vals, dtype=dtype, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return t.tolist()

    def run_leg(frame_spp, steps, warmup):
        """Builds the scene at `frame_spp`, renders warmup + steps frames; returns (scene, tile_buf, stats dict)."""
        if args.multi_handle:   # one handle over the N GPUs (rehearsal: N contexts on GPU 0)
            scene = pbrt_hip.Scene(devices=[0] * n_gpus if args.backend == "gloo" else list(range(n_gpus)))
        else:
            scene = pbrt_hip.Scene(device=local_rank)
        t_setup = time.time()
        capture(scene, frame_spp, device_build=not args.host_build)
        t_setup = time.time() - t_setup
        floats = max(scene.tile_buffer_floats(tile_size, p, world) for p in range(world))
        tile_buf = torch.zeros(floats, dtype=torch.float32, device=dev)
        gather_list = [torch.zeros(floats, dtype=torch.float32, device=dev) for _ in range(world)] if (use_dist and rank == 0) else None
        # the film comes back into page-locked host arrays allocated once (67 MB at 2048 x 2048: a DMA transfer instead of a staged copy into freshly faulted pages every frame)
        film_out = None
        if rank == 0:
            fh, fw = scene.film_shape
            film_out = (torch.empty((fh, fw, 3), dtype=torch.float32, pin_memory=True).numpy(), torch.empty((fh, fw), dtype=torch.float32, pin_memory=True).numpy())
        torch.cuda.synchronize()  # the library writes tile_buf on its own stream: torch's fill kernels must have finished

        stage = {"render": 0.0, "gather_merge": 0.0}   # host clock per stage of this rank's steps (the render call returns after its stream has drained)

        def step():
            ts = time.perf_counter()
            if args.multi_handle:   # tiles dealt to the handle's devices, gathered and merged inside the library
                xyz, wt, st = scene.render_path(max_depth=max_depth, tile_size=tile_size, out=film_out)
                stage["render"] += time.perf_counter() - ts
                return st, (xyz, wt)
            st = scene.render_path_tiles_device(tile_buf.data_ptr(), max_depth=max_depth, tile_size=tile_size, tile_part=rank, tile_parts=world)
            tr = time.perf_counter()
            stage["render"] += tr - ts
            film = step_tail()
            stage["gather_merge"] += time.perf_counter() - tr
            return st, film

        def step_tail():
            film = None
            if use_dist:
                if args.backend == "nccl":
                    dist.gather(tile_buf, gather_list, dst=0)   # RCCL over xGMI, device buffers
                else:
                    host_parts = [torch.zeros(floats) for _ in range(world)] if rank == 0 else None
                    dist.gather(tile_buf.cpu(), host_parts, dst=0)
                    if rank == 0:
                        for g, h in zip(gather_list, host_parts):
                            g.copy_(h)
                torch.cuda.synchronize()  # the collective runs on torch's stream: finish it before the library reads (rank 0) or rewrites (all) the buffers
                if rank == 0:
                    film = scene.merge_tiles_device([t.data_ptr() for t in gather_list], tile_size, out=film_out)
            else:
                film = scene.merge_tiles_device([tile_buf.data_ptr()], tile_size, out=film_out)
            return film

        for _ in range(warmup):
            step()
        sync()
        stage["render"] = stage["gather_merge"] = 0.0
        t0 = time.perf_counter()
        rays = reg = shd = launches = 0
        ext_s = sh_s = shade_s = 0.0
        film = None
        for _ in range(steps):
            st, film = step()
            reg += st.regular_rays; shd += st.shadow_rays
            ext_s += st.extend_seconds; sh_s += st.shadow_seconds; shade_s += st.shade_seconds
            launches = int(st.extend_launches + st.shadow_launches)
        sync()
        elapsed = time.perf_counter() - t0
        rays = reg + shd
        per_rank = None
        if use_dist:
            # every rank's own clock, render time and ray count (what the first real N-GPU line needs to be diagnosable): gathered, not only reduced
            mine = torch.tensor([elapsed, stage["render"], stage["gather_merge"], float(rays)], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(allv, mine)
            per_rank = [[float(x) for x in v.tolist()] for v in allv]
            elapsed = float(reduce_scalars([elapsed], dist.ReduceOp.MAX, torch.float64)[0])
            rays, reg, shd = (int(v) for v in reduce_scalars([rays, reg, shd], dist.ReduceOp.SUM, torch.int64))
        return scene, tile_buf, dict(elapsed=elapsed, rays=rays, reg=reg, shd=shd, ext_s=ext_s, sh_s=sh_s, shade_s=shade_s, launches=launches, film=film,
                                     t_setup=t_setup, steps=steps, stage=dict(stage), per_rank=per_rank)

    scaling = args.scaling if n_gpus > 1 else "weak"  # one GPU: the two coincide; the contract's default label
    frame_spp = spp * (n_gpus if (n_gpus > 1 and scaling == "weak") else 1)
    scene, tile_buf, r = run_leg(frame_spp, args.steps, args.warmup)

    out = None
    if rank == 0:
        mrays = r["rays"] / r["elapsed"] / 1e6
        label = cfg["label"] if not custom else f"custom ({', '.join(custom)} overridden; base {cfg['label']})"
        spp_note = f" (= {spp} x {n_gpus} GPUs, weak scaling)" if frame_spp != spp else ""
        acc = scene.accel_stats()
        out = {
            "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(r["elapsed"] / args.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("" if not args.instances else f"INSTANCED x{args.instances} (one object, two-level BVH) — ") + label +
                                   (f" — as run: {n_tris} random triangles (seed {args.seed}), single SAH BVH (maxnodeprims 4), {res}x{yres} @ {frame_spp} spp{spp_note}, "
                                    f"PathIntegrator maxdepth {max_depth}, halton, box filter, constant infinite light, " + ("matte Kd 0.5" if args.material == "matte" else f"material {args.material}")
                                    if sm is None else
                                    f" — as run: {json.dumps(sm.counts())}, {sm.n_materials} materials, two-level SAH BVH (maxnodeprims 4), {res}x{yres} @ {frame_spp} spp{spp_note}, "
                                    f"PathIntegrator maxdepth {max_depth}, halton, box filter, lightsamplestrategy spatial"),
                       "baseline_config": cfg_name if not custom else "custom", "key": workload_key(n_tris, res, frame_spp, max_depth, args.seed, yres, args.instances),
                       "tiles": (f"16x16, tile t on device t % {n_gpus} of ONE multi-device handle (one host thread per device), film tiles gathered on the first device inside the library"
                                 + (" (RCCL send / recv)" if args.backend == "nccl" else " (device-to-device copies: the contexts share GPU 0)")) if args.multi_handle else
                                f"16x16, tile t on rank t % {world}, film tiles gathered on rank 0 (" + ("RCCL" if args.backend == "nccl" else "gloo, through host memory: rehearsal on one GPU") + ")" if world > 1 else "16x16, one rank",
                       "rays_per_frame": r["rays"] // args.steps, "regular_rays_per_frame": r["reg"] // args.steps, "shadow_rays_per_frame": r["shd"] // args.steps,
                       "scene_setup_seconds_host": round(r["t_setup"], 3),
                       "accel": {"interior_nodes": acc["interior_nodes"], "leaf_records": acc["leaf_records"],
                                 "resident_bytes": acc["node_bytes"] + acc["leaf_record_bytes"], "build_seconds": round(acc["build_seconds"], 3), "built_on": "host" if args.host_build else "device"}},
            "film_sha256": hashlib.sha256(r["film"][0].tobytes() + r["film"][1].tobytes()).hexdigest(),
            "stage_ms_per_step_rank0": {"traversal": round((r["ext_s"] + r["sh_s"]) / args.steps * 1e3, 3),
                                        "raygen_shade_film": round(r["shade_s"] / args.steps * 1e3, 3)},
            # host clock around the two halves of a step on rank 0: the render call (returns after its stream drained) and what follows it (N > 1: the gather of the
            # film-tile buffers + the merge in tile order + film read-back; N = 1: merge + read-back)
            "stage_ms": {"render": round(r["stage"]["render"] / args.steps * 1e3, 3), "gather_merge": round(r["stage"]["gather_merge"] / args.steps * 1e3, 3)},
        }
        if r["per_rank"]:
            rm = [v[1] / args.steps * 1e3 for v in r["per_rank"]]
            out["ranks"] = {"render_ms": {"min": round(min(rm), 3), "mean": round(sum(rm) / len(rm), 3), "max": round(max(rm), 3), "per_rank": [round(v, 3) for v in rm]},
                            "imbalance": round(max(rm) / (sum(rm) / len(rm)), 4),
                            "gather_merge_ms_per_rank": [round(v[2] / args.steps * 1e3, 3) for v in r["per_rank"]],
                            "rays_per_rank": [int(v[3]) // args.steps for v in r["per_rank"]],
                            "elapsed_s_per_rank": [round(v[0], 4) for v in r["per_rank"]]}
            out["rccl"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank": dist.get_rank()}

    # ---- roofline of the dominant kernel (BVH traversal, both ray kinds), rank 0's share ------------------------------------------------
    if rank == 0:
        resident = acc["node_bytes"] + acc["leaf_record_bytes"]
        roof = {"bound": "hbm" if resident > INFINITY_CACHE_BYTES else "l2+infinity-cache",
                "bound_note": (f"BVH nodes + leaf records = {resident / 2**20:.0f} MiB " + ("exceed" if resident > INFINITY_CACHE_BYTES else "fit in") + " the 256 MiB Infinity Cache"),
                "kernel": "ph::traverse_kernel (BVH traversal + triangle tests; one launch per wavefront round serves the round's closest-hit and any-hit rays)",
                "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
        if not args.no_roofline_count:
            scene.set_traversal_counting(True)
            # (a multi-device handle's *_tiles_device entry points act on its first device: that device's share of the frame)
            scene.render_path_tiles_device(tile_buf.data_ptr(), max_depth=max_depth, tile_size=tile_size, tile_part=rank, tile_parts=n_gpus)
            cnt = scene.traversal_counts()
            scene.set_traversal_counting(False)
            cl, ah = cnt["closest"], cnt["any_hit"]
            # 32 B per reference-format node visit + 48 B per triangle test + ray in / result out (32 + 32 closest hit, 32 + 1 any hit)
            bytes_cl = 32 * cl["ref_node_visits"] + 48 * cl["tri_tests"] + 64 * cl["rays"]
            bytes_ah = 32 * ah["ref_node_visits"] + 48 * ah["tri_tests"] + 33 * ah["rays"]
            bytes_per_frame = bytes_cl + bytes_ah
            n_rays = cl["rays"] + ah["rays"]
            trav_per_frame = (r["ext_s"] + r["sh_s"]) / args.steps
            launches = r["launches"]
            if trav_per_frame > 0 and launches > 0:
                ach = bytes_per_frame / trav_per_frame / 1e9
                avg_launch_s = trav_per_frame / launches
                roof.update({"achieved": round(ach, 1), "frac_algorithmic": round(ach / HBM_PEAK_GBS, 4),
                             "frac_algorithmic_note": ("> 1: served from L2, not an HBM fraction" if ach / HBM_PEAK_GBS > 1.0 else
                                                       "algorithmic bytes over time over the HBM peak: an upper bound on the HBM fraction only if every node visit missed the caches; see `frac` (counters) and `bound`"),
                             "algorithmic_bytes_per_launch": bytes_per_frame // launches, "launches_per_step": launches,
                             "avg_launch_ms": round(avg_launch_s * 1e3, 4), "rays_per_step": n_rays,
                             "closest_hit": {"rays": cl["rays"], "ref_node_visits_per_ray": round(cl["ref_node_visits"] / max(cl["rays"], 1), 2),
                                             "tri_tests_per_ray": round(cl["tri_tests"] / max(cl["rays"], 1), 2), "bytes_per_ray": round(bytes_cl / max(cl["rays"], 1), 1)},
                             "any_hit": {"rays": ah["rays"], "ref_node_visits_per_ray": round(ah["ref_node_visits"] / max(ah["rays"], 1), 2),
                                         "tri_tests_per_ray": round(ah["tri_tests"] / max(ah["rays"], 1), 2), "bytes_per_ray": round(bytes_ah / max(ah["rays"], 1), 1)},
                             "kernel_Mrays_per_s": round(n_rays / trav_per_frame / 1e6, 1),
                             "node_steps_per_ray": round((cl["nodes_passed"] + ah["nodes_passed"]) / max(n_rays, 1), 2),
                             "note": "achieved = algorithmic bytes (SURVEY 8d, reference-format 32-B nodes, counted by an untimed counting pass of the same frame) / HIP-event time "
                                     "of every traversal launch of the timed steps; frac = memory-side PMC bytes / that time / peak when a PMC run of this workload is on file"})
                # memory-side traffic of the same kernel from rocprofv3 PMC passes (separate runs of this script under `rocprofv3 --pmc`, scripts/pmc_profile.sh;
                # MI355X_MICROARCH.md: FETCH_SIZE doubles for wide coalesced streaming reads only — this kernel reads random 64-B lines, for which the
                # calibration run scripts/calib/fetch_calib.hip shows FETCH_SIZE exact, so fetch_scale is 1).  Only a run of this exact workload counts.
                plain = n_gpus == 1 and args.material == "matte"
                e, why = find_traffic(workload_key(n_tris, res, frame_spp, max_depth, args.seed, yres, args.instances)) if plain else (None, "PMC records are kept for the single-GPU BASELINE workloads only")
                if e is not None:
                    k = e["traversal"]
                    per_launch = (float(e.get("fetch_scale", 1.0)) * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0 / k["dispatches"]
                    mem_gbs = per_launch / avg_launch_s / 1e9
                    l2_hit = k["TCC_HIT"] / (k["TCC_HIT"] + k["TCC_MISS"]) if k.get("TCC_HIT") and k.get("TCC_MISS") else None
                    hbm_frac = mem_gbs / HBM_PEAK_GBS
                    # what binds, from the counters alone: a launch whose requests are served by L2 and that draws a small share of the HBM peak is paced by the
                    # latency of its dependent L2 hits, not by memory bandwidth
                    if l2_hit is not None and l2_hit > 0.8 and hbm_frac < 0.2:
                        roof["bound"] = "l2-latency"
                        roof["bound_note"] = (f"derived from the PMC record: L2 hit rate {l2_hit:.3f} > 0.8 and memory-side traffic = {hbm_frac:.3f} of the HBM peak < 0.2 -> a chain of dependent "
                                              f"L2 hits; `frac` stays the HBM counter fraction, `l2` and `l1_frac_of_calibrated` place the kernel against the roofs that are nearer (" + roof["bound_note"] + ")")
                    elif hbm_frac >= 0.2:
                        roof["bound"] = "hbm"
                    # round 4: what the "l2-latency" label does and does not mean for THIS kernel (profiles/r04_quad_experiment/, r04_phase_clock_*): its data is L2-resident and every
                    # step waits on an L2 hit, but halving the number of dependent fetches per ray at equal fetch volume made it 11 % slower — the waves' time goes to instruction
                    # issue at a lane utilisation below one half, not to the length of the chain
                    roof["bound_evidence"] = ("round-4 experiment: two tree levels per fetch halve the dependent chain (58.4 -> 29.4 steps per ray, same L1 look-ups) and cost 11 % — the chain's length "
                                              "does not pace the kernel; issue slots at the lane utilisation in `valu` do (profiles/r04_quad_experiment/README.txt, profiles/r04_phase_clock_config2_32spp.txt)")
                    sqk = k.get("sq") or {}
                    if sqk.get("tcp_tcc_read_req"):
                        l2_bytes = sqk["tcp_tcc_read_req"] * 64.0 / k["dispatches"]
                        roof["l2"] = {"bytes": int(l2_bytes), "GBs": round(l2_bytes / avg_launch_s / 1e9, 1), "peak_GBs": L2_PEAK_GBS,
                                      "frac_of_34.5TBs": round(l2_bytes / avg_launch_s / 1e9 / L2_PEAK_GBS, 4),
                                      "note": "TCP_TCC_READ_REQ x 64 B per launch / avg launch time, against the 8 XCDs' aggregate L2 read bandwidth"}
                    if sqk.get("l1_accesses"):
                        roof["l1_frac_of_calibrated"] = round(sqk["l1_accesses"] / k["dispatches"] / avg_launch_s / 1e9 / L1_LOOKUP_PEAK_G, 4)
                    roof.update({"traffic": int(per_launch), "achieved_memory_side": round(mem_gbs, 1), "frac": round(mem_gbs / HBM_PEAK_GBS, 4),
                                 "frac_basis": "memory-side: traffic / avg_launch_ms / peak",
                                 "l2_hit_rate": round(l2_hit, 4) if l2_hit is not None else None,
                                 "pmc": k.get("sq"),
                                 "limiter": (("no single pipe is saturated: of the chip's cycles the vector ALUs issue in %.0f %% (lane utilisation %.2f), the scalar units take %.2f instructions per CU-cycle, "
                                              "the L1 tag pipes %.2f look-ups per CU-cycle (each lane's 64-B node costs four); L2 serves %.0f %% of the requests since the rays of a launch are binned by origin cell.  "
                                              "Measured cuts — 16 %% fewer vector instructions, 11 %% fewer L1 look-ups, a seventh wave per SIMD — each left the time unchanged on the same box; the rate follows the resident waves only up to there "
                                              "(3 -> 4 -> 5 -> 6 blocks per CU: +21, +13, +9 %%) and the locality of a launch's rays (unbinned queues: -20 %%): a walk of dependent 64-B fetches at ~70 %% of what the L1 / L2 path sustains "
                                              "for this access pattern (HISTORY §4; DESIGN §4.1 for round 4's reading)") %
                                             (100.0 * k["sq"]["valu_issue_of_simd_quads"], k["sq"]["valu_lane_utilisation"], k["sq"]["scalar_and_branch_per_cu_cycle"], k["sq"]["l1_accesses_per_cu_cycle"],
                                              100.0 * k["TCC_HIT"] / (k["TCC_HIT"] + k["TCC_MISS"]))) if k.get("sq") and k["sq"].get("valu_issue_of_simd_quads") is not None else None,
                                 "pipes": ({"valu_issue": k["sq"]["valu_issue_of_simd_quads"], "valu_lane_util": k["sq"]["valu_lane_utilisation"],
                                            "scalar_and_branch_insts_per_cu_cycle": k["sq"]["scalar_and_branch_per_cu_cycle"],
                                            "l1_lookups_per_cu_cycle": k["sq"]["l1_accesses_per_cu_cycle"],
                                            "l1_lookups_G_per_s": round(k["sq"]["l1_accesses"] / k["dispatches"] / avg_launch_s / 1e9, 1),
                                            "l1_lookups_G_per_s_calibrated_peak": L1_LOOKUP_PEAK_G,
                                            "l1_frac_of_calibrated_peak": round(k["sq"]["l1_accesses"] / k["dispatches"] / avg_launch_s / 1e9 / L1_LOOKUP_PEAK_G, 4),
                                            "note": "PMC: SQ_ACTIVE_INST_VALU, SQ_INSTS_SALU + SQ_INSTS_BRANCH, TCP_TOTAL_CACHE_ACCESSES over GRBM_GUI_ACTIVE (sum of the 8 XCDs; x 32 = CU-cycles = SIMD quads); "
                                                    "the L1 peak is what scripts/calib/node_fetch.hip reaches with this kernel's access pattern (one random 64-B record per lane, 4 x dwordx4, L2-resident table): "
                                                    "220 G records/s = 880 G look-ups/s, profiles/r03_calib_node_fetch.txt"}
                                           if k.get("sq") and k["sq"].get("valu_issue_of_simd_quads") is not None else None),
                                 "valu": ({"issue_frac": k["sq"]["valu_issue_of_simd_quads"], "lane_util": k["sq"]["valu_lane_utilisation"],
                                           "effective": round(k["sq"]["valu_issue_of_simd_quads"] * k["sq"]["valu_lane_utilisation"], 4),
                                           "note": "share of the SIMDs' issue quads spent on VALU instructions x share of the 64 lanes those instructions use; round 3 removed a sixth of these instructions without effect, round 4 "
                                                   "halved the dependent chain without effect and gained from every cut in idle-lane time (bound_evidence): issue slots at this lane utilisation are where the waves' time goes"}
                                          if k.get("sq") and k["sq"].get("valu_issue_of_simd_quads") is not None else None),
                                 "traffic_note": "bytes per launch leaving the L2s (rocprofv3 PMC FETCH_SIZE + WRITE_SIZE over " + str(k["dispatches"]) + " traversal launches of one frame, " +
                                                 e.get("source", "profiles/") + "; " + e.get("calibration", "") + ")"})
                else:
                    roof.update({"frac": None, "frac_basis": "none: no PMC record of this workload (" + why + "); frac_algorithmic is not an HBM fraction", "traffic_error": why})
        out["roofline"] = roof

    # ---- CPU baseline: the oracle (port of the reference algorithm) on this box's host cores, bounded sample ------------------------
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_binding import OracleScene
        cores = host_cores()
        rays_per_spp = max(1, (r["rays"] // args.steps) // frame_spp)
        cpu_spp = args.cpu_spp if args.cpu_spp > 0 else max(1, min(spp, round(150e6 / rays_per_spp)))
        orc = OracleScene()
        tb = time.time()
        capture(orc, cpu_spp, device_build=False)
        tb = time.time() - tb
        _, _, ost, _ = orc.render_path_ex(max_depth=max_depth, threads=cores)
        crays = ost.regular_rays + ost.shadow_rays
        out["cpu_baseline"] = {"value": round(crays / ost.render_seconds / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                               "sample": f"same scene and resolution at {cpu_spp} spp (the first {cpu_spp} Halton samples of every pixel): {crays} rays in "
                                         f"{ost.render_seconds:.2f} s render phase; single-threaded SAH build {tb:.2f} s excluded",
                               "gpu_over_cpu": round(out["value"] / (crays / ost.render_seconds / 1e6), 1)}
        orc.close()

    # ---- N > 1, strong scaling: a short weak-scaling leg alongside (spp x N: every rank traces what one GPU traces at N = 1) ----------------
    if n_gpus > 1 and scaling == "strong" and not args.no_weak_leg:
        scene.close()
        del tile_buf
        torch.cuda.empty_cache()
        wsteps = max(1, min(args.steps, 3))
        scene, tile_buf, rw = run_leg(spp * n_gpus, wsteps, 1)
        if rank == 0:
            out["weak_alongside"] = {"value": round(rw["rays"] / rw["elapsed"] / 1e6, 2), "unit": "Mrays/s", "steps": wsteps, "warmup": 1,
                                     "ms_per_step": round(rw["elapsed"] / wsteps * 1e3, 3), "spp": spp * n_gpus,
                                     "note": "same scene and resolution at spp x N: per-GPU work equals the N = 1 frame's"}

    if rank == 0:
        print(json.dumps(out), flush=True)
    scene.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
