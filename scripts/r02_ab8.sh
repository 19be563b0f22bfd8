#!/bin/bash
# usage bash scripts/r02_ab8.sh <tag> <configs...>: paths per chunk (PBRT_HIP_MAX_PATHS) on the big configs
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
for CFG in "$@"; do for MP in ${MPS:-33554432 67108864 134217728 16777216}; do
  PBRT_HIP_MAX_PATHS=$MP python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-count > $O/c${CFG}_$MP.json 2> $O/c${CFG}_$MP.err || { echo "config $CFG max_paths $MP FAILED"; tail -3 $O/c${CFG}_$MP.err; continue; }
  python3 - <<PY
import json
d=json.loads([l for l in open('$O/c${CFG}_$MP.json').read().splitlines() if l.startswith('{')][-1])
print('config $CFG max_paths $MP', d['value'], d['ms_per_step'], d['stage_ms_per_step_rank0'], d['film_sha256'][:12])
PY
done; done
