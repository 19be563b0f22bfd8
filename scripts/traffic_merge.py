#!/usr/bin/env python3
"""Adds (or replaces) one PMC record in profiles/r04_traffic.json.  The record is tied to the library build it was measured on (sha256 of libpbrt_hip.so) and to the tuning
environment (PBRT_HIP_* variables): bench.py reports a memory-side roofline fraction only from a record that matches the library it is running.
usage: scripts/traffic_merge.py <gpurun_out/pmc_TAG/traffic.json> <source label, e.g. profiles/r03_pmc_summary_config2.txt> [waves per SIMD of the traversal kernel]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "r04_traffic.json")
entry = json.load(open(sys.argv[1]))
entry["source"] = sys.argv[2]
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (library_identity: the hash of the library's sources + the tuning environment)
entry["lib_sha256"], entry["env"] = bench.library_identity()
if len(sys.argv) > 3:
    entry["waves_per_simd"] = float(sys.argv[3])   # resident waves per SIMD of the traversal kernel that ran (6 flat, 4 instanced)
if "workload" not in entry:
    raise SystemExit("the PMC record names no workload (no pass*.json with a bench line next to it)")
doc = json.load(open(dst)) if os.path.exists(dst) else {"entries": []}
doc["entries"] = [e for e in doc["entries"] if e.get("workload") != entry["workload"]] + [entry]
json.dump(doc, open(dst, "w"), indent=1)
print("recorded", entry["workload"], "->", dst)
