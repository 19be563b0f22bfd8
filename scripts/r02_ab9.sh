#!/bin/bash
# usage bash scripts/r02_ab9.sh <tag> <config> VAR v1 v2 ...: one environment knob swept on one bench config
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; CFG=$2; VAR=$3; shift 3
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
for V in "$@"; do
  env $VAR=$V python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-count > $O/c${CFG}_${VAR}_$V.json 2> $O/c${CFG}_${VAR}_$V.err || { echo "config $CFG $VAR=$V FAILED"; tail -3 $O/c${CFG}_${VAR}_$V.err; continue; }
  python3 - <<PY
import json
d=json.loads([l for l in open('$O/c${CFG}_${VAR}_$V.json').read().splitlines() if l.startswith('{')][-1])
print('config $CFG $VAR=$V', d['value'], d['ms_per_step'], d['stage_ms_per_step_rank0'], d['film_sha256'][:12])
PY
done
