#!/bin/bash
# A/B of traversal-queue options on one box: usage bash scripts/r02_ab.sh <tag> <bench args...>
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
  run sort2_$CFG PBRT_HIP_SORT_RAYS=2
  run heads8_$CFG PBRT_HIP_TRAV_HEADS=8
  run sort1_heads8_$CFG PBRT_HIP_SORT_RAYS=1 PBRT_HIP_TRAV_HEADS=8
  run sort2_heads8_$CFG PBRT_HIP_SORT_RAYS=2 PBRT_HIP_TRAV_HEADS=8
  run occ4_$CFG PBRT_HIP_TRAV_BLOCKS_PER_CU=4
  run occ5_$CFG PBRT_HIP_TRAV_BLOCKS_PER_CU=5
done
./scripts/calib/gather_rate > $O/gather_rate.txt 2>&1; cat $O/gather_rate.txt
