#!/usr/bin/env python3
"""Adds (or replaces) one PMC record in profiles/r02_traffic.json.
usage: scripts/traffic_merge.py <gpurun_out/pmc_TAG/traffic.json> <source label, e.g. profiles/r02_pmc_summary_config2.txt>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "r02_traffic.json")
entry = json.load(open(sys.argv[1]))
entry["source"] = sys.argv[2]
if "workload" not in entry:
    raise SystemExit("the PMC record names no workload (no pass*.json with a bench line next to it)")
doc = json.load(open(dst)) if os.path.exists(dst) else {"entries": []}
doc["entries"] = [e for e in doc["entries"] if e.get("workload") != entry["workload"]] + [entry]
json.dump(doc, open(dst, "w"), indent=1)
print("recorded", entry["workload"], "->", dst)
