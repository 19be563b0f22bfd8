#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) into per-kernel sums."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    if k.startswith("__amd") or "at::native" in k:
        continue
    c = agg[k]
    print(f"== {k}  (dispatches: {max(calls[k].values())})")
    for name in sorted(c):
        print(f"   {name:40s} {c[name]:.6g}")
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        print(f"   -> VALU lane utilisation       {c['SQ_THREAD_CYCLES_VALU'] / (64.0 * c['SQ_ACTIVE_INST_VALU']):.3f}")
    if c.get("SQ_WAVE_CYCLES"):
        wc = c["SQ_WAVE_CYCLES"]
        print(f"   -> of wave cycles: wait_any {c.get('SQ_WAIT_ANY', 0) / wc:.3f}  wait_inst {c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}  valu_active {c.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}")
    if c.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles: 32 x it = CU-cycles of the chip's 256 CUs = SIMD-quads of its 1 024 SIMDs (SQ_ACTIVE_* count 4-cycle quads)
        cu = 32.0 * c["GRBM_GUI_ACTIVE"]
        print(f"   -> pipes: VALU issue {c.get('SQ_ACTIVE_INST_VALU', 0) / cu:.3f} of the SIMD quads   scalar + branch {(c.get('SQ_INSTS_SALU', 0) + c.get('SQ_INSTS_BRANCH', 0)) / cu:.3f} per CU-cycle   L1 (TCP) accesses {c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0) / cu:.3f} per CU-cycle")
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum"):
        print(f"   -> L2 hit rate                 {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
    if c.get("FETCH_SIZE"):
        print(f"   -> FETCH_SIZE KB (x2 for wide streaming reads on gfx950, MI355X_MICROARCH.md): {c['FETCH_SIZE']:.0f}")


# machine-readable summary of the memory-side counters: one entry of profiles/<round>_traffic.json (bench.py matches it by `workload`)
import json
out_json = {}
for k in agg:
    c = agg[k]
    if "FETCH_SIZE" in c:
        n = max(calls[k].values())
        out_json[k.strip()] = {"dispatches": n, "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c.get("WRITE_SIZE", 0.0),
                               "TCC_HIT": c.get("TCC_HIT_sum"), "TCC_MISS": c.get("TCC_MISS_sum")}
# all traversal launches of the frame together (round 0 = closest-hit only kernel, later rounds = the MIXED kernel)
tk = [k for k in out_json if "traverse_kernel" in k]
entry = {"kernels": out_json}
if tk:
    entry["traversal"] = {"kernels": tk, "dispatches": sum(out_json[k]["dispatches"] for k in tk),
                          "FETCH_SIZE_KB": sum(out_json[k]["FETCH_SIZE_KB"] for k in tk), "WRITE_SIZE_KB": sum(out_json[k]["WRITE_SIZE_KB"] for k in tk),
                          "TCC_HIT": sum(out_json[k]["TCC_HIT"] or 0 for k in tk), "TCC_MISS": sum(out_json[k]["TCC_MISS"] or 0 for k in tk)}
    # what the SIMDs were doing in those launches (same passes): the numbers behind "issue bound, not memory bound"
    sq = {n: sum(agg[k].get(n, 0.0) for k in agg if "traverse_kernel" in k) for n in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVES",
                                                                                    "SQ_INSTS_VALU", "SQ_INSTS_SALU", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "SQ_INSTS_BRANCH", "TCP_TOTAL_CACHE_ACCESSES_sum", "GRBM_GUI_ACTIVE",
                                                                                    "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS")}
    if sq["SQ_WAVE_CYCLES"]:
        entry["traversal"]["sq"] = {"valu_lane_utilisation": round(sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"]), 4) if sq["SQ_ACTIVE_INST_VALU"] else None,
                                    "valu_active_of_wave_cycles": round(sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"], 4), "wait_any_of_wave_cycles": round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 4),
                                    "wait_inst_of_wave_cycles": round(sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"], 4), "valu_insts": sq["SQ_INSTS_VALU"], "salu_insts": sq["SQ_INSTS_SALU"],
                                    "mean_l2_read_latency_cycles": round(sq["TCP_TCC_READ_REQ_LATENCY_sum"] / sq["TCP_TCC_READ_REQ_sum"], 1) if sq["TCP_TCC_READ_REQ_sum"] else None,
                                    # the three pipes the kernel loads: GRBM_GUI_ACTIVE sums the 8 XCDs, 32 x it = the chip's CU-cycles = its SIMD-quads
                                    "branch_insts": sq["SQ_INSTS_BRANCH"], "vmem_insts": sq["SQ_INSTS_VMEM_RD"] + sq["SQ_INSTS_VMEM_WR"], "lds_insts": sq["SQ_INSTS_LDS"],
                                    "l1_accesses": sq["TCP_TOTAL_CACHE_ACCESSES_sum"], "tcp_tcc_read_req": sq["TCP_TCC_READ_REQ_sum"], "gpu_busy_cycles_sum_over_xcds": sq["GRBM_GUI_ACTIVE"],
                                    "valu_issue_of_simd_quads": round(sq["SQ_ACTIVE_INST_VALU"] / (32.0 * sq["GRBM_GUI_ACTIVE"]), 4) if sq["GRBM_GUI_ACTIVE"] else None,
                                    "scalar_and_branch_per_cu_cycle": round((sq["SQ_INSTS_SALU"] + sq["SQ_INSTS_BRANCH"]) / (32.0 * sq["GRBM_GUI_ACTIVE"]), 4) if sq["GRBM_GUI_ACTIVE"] else None,
                                    "l1_accesses_per_cu_cycle": round(sq["TCP_TOTAL_CACHE_ACCESSES_sum"] / (32.0 * sq["GRBM_GUI_ACTIVE"]), 4) if sq["GRBM_GUI_ACTIVE"] else None}
# the workload the passes ran: every pass is one `bench.py --steps 1 --warmup 0` run whose JSON line was kept next to the counters
for f in sorted(glob.glob(os.path.join(out, "pass*.json"))):
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        entry["workload"] = json.loads(line)["config"]["key"]
        entry["workload_label"] = json.loads(line)["config"]["workload"]
        break
    except (IndexError, KeyError, ValueError):
        continue
entry["fetch_scale"] = 1.0
entry["calibration"] = ("FETCH_SIZE used unscaled: a calibration kernel with this kernel's access pattern (random 64-B records, 4 x dwordx4 per lane) reports 0.9995 of the "
                        "known bytes, profiles/r01_fetch_calibration.txt; the x2 of MI355X_MICROARCH.md applies to wide coalesced streaming reads")
with open(os.path.join(out, "traffic.json"), "w") as f:
    json.dump(entry, f, indent=1)
