#!/bin/bash
# rocprofv3 kernel trace + stats of one bench.py run:  bash scripts/kstats.sh TAG <bench.py arguments>
# writes gpurun_out/TAG_stats/ and prints the per-kernel summary (name, calls, total ms, average ms, share)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --no-roofline-count "$@" > $R/gpurun_out/${TAG}_stats.json 2> $R/gpurun_out/${TAG}_stats.err
cd $R
python3 - "$R/gpurun_out/${TAG}_stats" <<'PY'
import csv, glob, sys, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))
if not f: print("no kernel_stats.csv under", sys.argv[1]); sys.exit(1)
rows = list(csv.DictReader(open(f[-1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    name = re.sub(r"\(.*", "", r["Name"])[:110]
    print(f'{float(r["TotalDurationNs"])/1e6:10.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f} %  x{int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e6:9.3f} ms  {name}')
PY
