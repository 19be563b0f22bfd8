#!/bin/bash
# usage: scripts/variant_sweep.sh "0 1 2 ..." [bench args]  -> one line per variant
VARS=$1; shift
for v in $VARS; do
  PBRT_HIP_TRAV_VARIANT=$v python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-count "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('variant $v', 'Mrays/s', d['value'], 'ms', d['ms_per_step'], d['stage_ms_per_step_rank0'])"
done
