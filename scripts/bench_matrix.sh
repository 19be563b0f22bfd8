#!/bin/bash
# The secondary bench configurations whose JSON lines are kept under profiles/ (the headline ones come from scripts/r02_baseline.sh).
# usage (on the GPU box): bash scripts/bench_matrix.sh <outdir>
set -e
O=${1:-gpurun_out/matrix}; mkdir -p $O
for m in plastic glass metal uber mixed textured; do python bench.py --config 1 --material $m --steps 5 --warmup 2 --cpu-spp 16 > $O/bench_material_$m.json 2> $O/bench_material_$m.err; echo $m done; done
python bench.py --config 1 --instances 1000 --n-tris 10000 --steps 3 --warmup 1 --cpu-spp 4 > $O/bench_instanced_1000x10k.json 2> $O/bench_instanced.err
echo inst done
for f in $O/bench_*.json; do python3 - <<PY
import json
d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1])
print('$f'.split('/')[-1], d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['stage_ms_per_step_rank0'], 'x cpu', d.get('cpu_baseline',{}).get('gpu_over_cpu'))
PY
done
