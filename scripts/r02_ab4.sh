#!/bin/bash
# usage bash scripts/r02_ab4.sh <tag>: shade-kernel prefetch and instancing-kernel variants
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
run() {  # name, bench args (quoted), env...
  n=$1; a=$2; shift; shift
  env "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-count $a > $O/$n.json 2> $O/$n.err || { echo "$n FAILED"; tail -3 $O/$n.err; return; }
  python3 - <<PY
import json
d=json.loads([l for l in open('$O/$n.json').read().splitlines() if l.startswith('{')][-1])
print('$n', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['stage_ms_per_step_rank0'], d['film_sha256'][:12])
PY
}
for c in 2 1; do
  run c${c}_base "--config $c" A=0
  run c${c}_prefetch "--config $c" PBRT_HIP_SHADE_PREFETCH=1
done
for v in 0 1 2 3; do run inst_v$v "--config 1 --instances 1000 --n-tris 10000" PBRT_HIP_INST_VARIANT=$v; done
