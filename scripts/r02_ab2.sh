#!/bin/bash
# A/B of head chunk sizes with binned queues: usage bash scripts/r02_ab2.sh <tag> <configs...>
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
run() {  # name, env...
  n=$1; shift
  env "$@" python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline-count $ARGS > $O/$n.json 2> $O/$n.err || { echo "$n FAILED"; tail -3 $O/$n.err; return; }
  python3 - <<PY
import json
d=json.loads([l for l in open('$O/$n.json').read().splitlines() if l.startswith('{')][-1])
print('$n', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['stage_ms_per_step_rank0'], d['film_sha256'][:12])
PY
}
for CFG in "$@"; do
  ARGS="--config $CFG"
  echo "== config $CFG"
  run base_$CFG A=0
  run sort1_$CFG PBRT_HIP_SORT_RAYS=1
  for c in 2048 16384 49152 131072; do
    run sort1_h8_c${c}_$CFG PBRT_HIP_SORT_RAYS=1 PBRT_HIP_TRAV_HEADS=8 PBRT_HIP_HEAD_CHUNK=$c
  done
  run h8_c49152_$CFG PBRT_HIP_TRAV_HEADS=8 PBRT_HIP_HEAD_CHUNK=49152
done
