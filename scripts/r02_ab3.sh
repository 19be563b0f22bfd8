#!/bin/bash
# usage bash scripts/r02_ab3.sh <tag> <configs...>: the shipping defaults against the options switched off
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
run() {  # name, env...
  n=$1; shift
  env "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-count $ARGS > $O/$n.json 2> $O/$n.err || { echo "$n FAILED"; tail -3 $O/$n.err; return; }
  python3 - <<PY
import json
d=json.loads([l for l in open('$O/$n.json').read().splitlines() if l.startswith('{')][-1])
print('$n', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['stage_ms_per_step_rank0'], d['film_sha256'][:12])
PY
}
for CFG in "$@"; do
  ARGS="--config $CFG"
  echo "== config $CFG"
  run default_$CFG A=0
  run nosort_noheads_$CFG PBRT_HIP_SORT_RAYS=0 PBRT_HIP_TRAV_HEADS=1
  run sort2_$CFG PBRT_HIP_SORT_RAYS=2
done
