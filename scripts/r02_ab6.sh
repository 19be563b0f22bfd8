#!/bin/bash
# usage bash scripts/r02_ab6.sh <tag>: texture-pass variants on the textured configs[1]
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
for w in 4 3 2; do
  PBRT_HIP_TEX_WAVES=$w python3 bench.py --config 1 --material textured --steps 5 --warmup 2 --no-cpu-baseline --no-roofline-count > $O/tex_w$w.json 2> $O/tex_w$w.err || { echo "w$w FAILED"; tail -3 $O/tex_w$w.err; continue; }
  python3 - <<PY
import json
d=json.loads([l for l in open('$O/tex_w$w.json').read().splitlines() if l.startswith('{')][-1])
print('textured, texture pass compiled for $w waves', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['stage_ms_per_step_rank0'], d['film_sha256'][:12])
PY
done
