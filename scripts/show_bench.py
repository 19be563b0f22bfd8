#!/usr/bin/env python3
"""Prints the headline fields of bench.py JSON lines: scripts/show_bench.py <file> [<file> ...]"""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except (IndexError, OSError, ValueError) as e:
        print(f, "no bench line:", e)
        continue
    r = d.get("roofline") or {}
    print(f.split("/")[-1], d["value"], d["unit"], d["ms_per_step"], "ms", d["stage_ms_per_step_rank0"], "film", d["film_sha256"][:12], "bound", r.get("bound"), "frac", r.get("frac"),
          "cpu", (d.get("cpu_baseline") or {}).get("value"))
