#!/usr/bin/env python3
"""bench.py — Mrays/s of the MI355X wavefront path tracer on BASELINE.json's configs[1]
("Synthetic 100k random triangles, single BVH, 512x512 @ 64 spp, 1x MI355X").

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one frame: SamplerIntegrator::render of the whole image (raygen -> [extend, shadow, shade] x (maxdepth+1) -> film).
Scene, BVH and sampler tables are resident in HBM before the timed region.  With N ranks the frame's 16x16 tiles are dealt
round-robin (tile t -> rank t % N, the reference's tile enumeration), every rank renders its tiles, the per-tile film buffers
are gathered on rank 0 over RCCL (the path's one exchange step) and merged there in tile order.
Scaling (N > 1): "weak" by default — the frame is rendered at N x the samples per pixel, so every rank traces as many paths as
the single GPU does at N = 1 and the job is N frames' worth of work; `--scaling strong` renders the identical N = 1 frame
instead (same film bits as one rank, tests/test_multi_rank_gpu.py).

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel (BVH traversal; one launch per wavefront round serves that round's closest-hit
and any-hit rays): achieved = algorithmic bytes (32 B per reference-format node visit + 48 B per triangle test + 64 B ray in / hit
out, 33 B for any-hit rays) / its HIP-event time measured inside the timed steps.  `cpu_baseline` times the CPU oracle (oracle/, a port of the reference algorithm — the Rust
reference cannot be built here) on a bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pbrt-v3-rs_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def host_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota and by the box's per-GPU
    CPU share (16)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("PBRT_HIP_CPU_THREADS", "16"))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-tris", type=int, default=100_000)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-spp", type=int, default=64, help="spp of the bounded CPU-baseline sample (same scene, same resolution)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-count", action="store_true")
    ap.add_argument("--material", default="matte", choices=["matte", "plastic", "glass", "metal", "uber", "mixed", "textured"],
                    help="material of every triangle; anything but matte runs the general-BSDF shade kernel (not the headline workload)")
    ap.add_argument("--instances", type=int, default=0,
                    help="K > 0: the triangles become one object instanced K times (two-level BVH, TransformedPrimitive path); not the headline workload")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = spp x N (per-GPU work fixed), strong = the N=1 frame split over N ranks")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = functional rehearsal of the N>1 path on ONE GPU: all ranks share device 0 and film tiles travel through host memory")
    args = ap.parse_args()

    import numpy as np
    import torch
    import pbrt_hip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if args.backend == "gloo":
        local_rank = 0  # rehearsal mode: every rank drives GPU 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    host = pbrt_hip.Host()
    frame_spp = args.spp * (world if args.scaling == "weak" else 1)
    spec = pbrt_hip.SceneSpec(n_tris=args.n_tris, seed=args.seed, xres=args.res, yres=args.res, spp=frame_spp, max_depth=args.max_depth, material=args.material)
    scene = pbrt_hip.Scene(device=local_rank)
    t_setup = time.time()
    geometry = pbrt_hip.capture_spec(spec, scene, host, instances=args.instances)
    t_setup = time.time() - t_setup

    tile_size = 16
    floats = max(scene.tile_buffer_floats(tile_size, p, world) for p in range(world))
    tile_buf = torch.zeros(floats, dtype=torch.float32, device=dev)
    gather_list = [torch.zeros(floats, dtype=torch.float32, device=dev) for _ in range(world)] if (world > 1 and rank == 0) else None
    torch.cuda.synchronize()  # the library writes tile_buf on its own stream: torch's fill kernels must have finished

    def step():
        st = scene.render_path_tiles_device(tile_buf.data_ptr(), max_depth=args.max_depth, tile_size=tile_size, tile_part=rank, tile_parts=world)
        film = None
        if world > 1:
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
                film = scene.merge_tiles_device([t.data_ptr() for t in gather_list], tile_size)
        else:
            film = scene.merge_tiles_device([tile_buf.data_ptr()], tile_size)
        return st, film

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_scalars(vals, op, dtype):
        t = torch.tensor(vals, dtype=dtype, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return t.tolist()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    rays = 0
    ext_s = sh_s = shade_s = 0.0
    reg = shd = 0
    film = None
    for _ in range(args.steps):
        st, film = step()
        rays += st.regular_rays + st.shadow_rays
        reg += st.regular_rays; shd += st.shadow_rays
        ext_s += st.extend_seconds; sh_s += st.shadow_seconds; shade_s += st.shade_seconds
    sync()
    elapsed = time.perf_counter() - t0

    if world > 1:
        elapsed = float(reduce_scalars([elapsed], dist.ReduceOp.MAX, torch.float64)[0])
        rays, reg, shd = (int(v) for v in reduce_scalars([rays, reg, shd], dist.ReduceOp.SUM, torch.int64))

    out = None
    spp_note = f" (= {args.spp} x {world} ranks)" if frame_spp != args.spp else ""
    if rank == 0:
        mrays = rays / elapsed / 1e6
        out = {
            "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("" if not args.instances else f"INSTANCED x{args.instances} (one object, two-level BVH) — ") +
                                   f"configs[1]: {args.n_tris} random triangles (seed {args.seed}), single SAH BVH, {args.res}x{args.res} @ {frame_spp} spp{spp_note}, "
                                   f"PathIntegrator maxdepth {args.max_depth}, halton, box filter, constant infinite light, " + ("matte Kd 0.5" if args.material == "matte" else f"material {args.material}"),
                       "tiles": "16x16, tile t on rank t % n_gpus, film tiles gathered on rank 0 (RCCL)" if world > 1 else "16x16, one rank",
                       "rays_per_frame": rays // args.steps, "regular_rays_per_frame": reg // args.steps, "shadow_rays_per_frame": shd // args.steps,
                       "scene_setup_seconds_host": round(t_setup, 3)},
            "film_sha256": __import__("hashlib").sha256(film[0].tobytes() + film[1].tobytes()).hexdigest(),
            "stage_ms_per_step_rank0": {"extend": round(ext_s / args.steps * 1e3, 3), "shadow": round(sh_s / args.steps * 1e3, 3),
                                        "raygen_shade_film": round(shade_s / args.steps * 1e3, 3)},
        }

    # ---- roofline of the dominant kernel (BVH traversal, both ray kinds), rank 0's share ------------------------------------------------
    if rank == 0:
        roof = {"bound": "hbm", "kernel": "ph::traverse_kernel (BVH traversal + triangle tests; one launch per wavefront round serves the round's closest-hit and any-hit rays)", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": None, "traffic": None}
        if not args.no_roofline_count:
            scene.set_traversal_counting(True)
            st_c = scene.render_path_tiles_device(tile_buf.data_ptr(), max_depth=args.max_depth, tile_size=tile_size, tile_part=rank, tile_parts=world)
            cnt = scene.traversal_counts()
            scene.set_traversal_counting(False)
            cl, ah = cnt["closest"], cnt["any_hit"]
            # 32 B per reference-format node visit + 48 B per triangle test + ray in / result out (32 + 32 closest hit, 32 + 1 any hit)
            bytes_cl = 32 * cl["ref_node_visits"] + 48 * cl["tri_tests"] + 64 * cl["rays"]
            bytes_ah = 32 * ah["ref_node_visits"] + 48 * ah["tri_tests"] + 33 * ah["rays"]
            bytes_per_frame = bytes_cl + bytes_ah
            n_rays = cl["rays"] + ah["rays"]
            trav_per_frame = (ext_s + sh_s) / args.steps
            launches = int(st.extend_launches + st.shadow_launches)
            ach = bytes_per_frame / trav_per_frame / 1e9 if trav_per_frame > 0 else None
            roof.update({"achieved": round(ach, 1) if ach else None, "frac": round(ach / HBM_PEAK_GBS, 4) if ach else None,
                         "algorithmic_bytes_per_launch": bytes_per_frame // max(launches, 1), "launches_per_step": launches,
                         "avg_launch_ms": round(trav_per_frame / launches * 1e3, 4), "rays_per_step": n_rays,
                         "closest_hit": {"rays": cl["rays"], "ref_node_visits_per_ray": round(cl["ref_node_visits"] / max(cl["rays"], 1), 2),
                                         "tri_tests_per_ray": round(cl["tri_tests"] / max(cl["rays"], 1), 2), "bytes_per_ray": round(bytes_cl / max(cl["rays"], 1), 1)},
                         "any_hit": {"rays": ah["rays"], "ref_node_visits_per_ray": round(ah["ref_node_visits"] / max(ah["rays"], 1), 2),
                                     "tri_tests_per_ray": round(ah["tri_tests"] / max(ah["rays"], 1), 2), "bytes_per_ray": round(bytes_ah / max(ah["rays"], 1), 1)},
                         "kernel_Mrays_per_s": round(n_rays / trav_per_frame / 1e6, 1) if trav_per_frame > 0 else None,
                         "note": "counts from an untimed counting pass of the same frame (node visits and triangle tests the reference's traversal makes for these rays); "
                                 "time = HIP events around every traversal launch of the timed steps"})
        # HBM-side traffic of the same kernel from rocprofv3 PMC passes (separate runs of this script under `rocprofv3 --pmc`,
        # scripts/pmc_profile.sh; summary committed under profiles/).  MI355X_MICROARCH.md prescribes doubling FETCH_SIZE for wide coalesced
        # streaming reads; this kernel's reads are random 64-B lines (4 x dwordx4 per lane), for which a calibration run on a known byte count
        # (scripts/calib/fetch_calib.hip, profiles/r01_fetch_calibration.txt) shows FETCH_SIZE to be exact — so it is used unscaled.
        # Only used when it was collected on this exact workload.
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and roof.get("achieved"):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == [args.n_tris, args.res, args.spp, args.max_depth, args.seed] and world == 1 and args.material == "matte" and not args.instances:
                    k = tj["traversal"]
                    per_launch = (float(tj.get("fetch_scale", 1.0)) * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0 / k["dispatches"]
                    roof["traffic"] = int(per_launch)
                    roof["traffic_note"] = ("HBM-side bytes per launch from rocprofv3 PMC (FETCH_SIZE + WRITE_SIZE, " + tj.get("source", "profiles/") + "; FETCH_SIZE calibrated exact "
                                            "for this access pattern, " + tj.get("calibration", "") + "); algorithmic bytes are counted in the reference's 32-B-node format, "
                                            "most of them are served by L2/MALL (L2 hit %.3f)" % (k["TCC_HIT"] / (k["TCC_HIT"] + k["TCC_MISS"])))
            except Exception:
                pass
        out["roofline"] = roof

    # ---- CPU baseline: the oracle (port of the reference algorithm) on this box's host cores, bounded sample ------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_binding import OracleScene
        cores = host_cores()
        cspec = pbrt_hip.SceneSpec(n_tris=args.n_tris, seed=args.seed, xres=args.res, yres=args.res, spp=args.cpu_spp, max_depth=args.max_depth, material=args.material)
        orc = OracleScene()
        tb = time.time()
        pbrt_hip.capture_spec(cspec, orc, host, geometry=geometry, instances=args.instances)
        tb = time.time() - tb
        _, _, ost, _ = orc.render_path_ex(max_depth=args.max_depth, threads=cores)
        crays = ost.regular_rays + ost.shadow_rays
        out["cpu_baseline"] = {"value": round(crays / ost.render_seconds / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                               "sample": f"same scene and resolution at {args.cpu_spp} spp (the first {args.cpu_spp} Halton samples of every pixel): {crays} rays in "
                                         f"{ost.render_seconds:.2f} s render phase; single-threaded SAH build {tb:.2f} s excluded",
                               "gpu_over_cpu": round(out["value"] / (crays / ost.render_seconds / 1e6), 1)}
        orc.close()

    if rank == 0:
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
