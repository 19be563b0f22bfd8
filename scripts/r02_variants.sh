#!/bin/bash
# traversal-loop tuning variants (api.hip PH_VARIANTS) with binned queues: usage bash scripts/r02_variants.sh <tag> <configs...>
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
for CFG in "$@"; do
  echo "== config $CFG"
  for v in ${VARIANTS:-0 1 2 3 4 5 6 7 8 9}; do
    PBRT_HIP_TRAV_VARIANT=$v python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-count > $O/v${v}_$CFG.json 2> $O/v${v}_$CFG.err || { echo "v$v FAILED"; continue; }
    python3 - <<PY
import json
d=json.loads([l for l in open('$O/v${v}_$CFG.json').read().splitlines() if l.startswith('{')][-1])
print('variant $v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['stage_ms_per_step_rank0'], d['film_sha256'][:12])
PY
  done
done
