#!/bin/bash
# usage: scripts/env_sweep.sh VAR "v1 v2 ..." [bench args]
VAR=$1; VALS=$2; shift; shift
for v in $VALS; do
  env $VAR=$v python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-count "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$VAR=$v', 'Mrays/s', d['value'], 'ms', d['ms_per_step'], d['stage_ms_per_step_rank0'])"
done
