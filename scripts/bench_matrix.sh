#!/bin/bash
# The secondary bench configurations whose JSON lines are kept under profiles/ (the headline one is scripts/profile_round.sh).
# usage (on the GPU box): bash scripts/bench_matrix.sh <outdir>
set -e
O=${1:-gpurun_out/matrix}; mkdir -p $O
python bench.py --n-tris 1000000 --steps 3 --warmup 1 --cpu-spp 16 > $O/bench_1Mtris.json 2> $O/bench_1Mtris.err
echo 1M done
python bench.py --n-tris 10000000 --res 2048 --spp 16 --steps 2 --warmup 1 --cpu-spp 1 > $O/bench_10Mtris_2048.json 2> $O/bench_10Mtris_2048.err
echo 10M done
python bench.py --n-tris 4300000 --res 1024 --spp 256 --max-depth 8 --steps 1 --warmup 1 --cpu-spp 1 > $O/bench_4p3Mtris_1024_256spp_depth8.json 2> $O/bench_4p3M.err
echo 4.3M done
for m in plastic glass metal uber mixed textured; do python bench.py --material $m --steps 5 --warmup 2 --cpu-spp 16 > $O/bench_material_$m.json 2> $O/bench_material_$m.err; echo $m done; done
python bench.py --instances 1000 --n-tris 10000 --steps 3 --warmup 1 --cpu-spp 4 > $O/bench_instanced_1000x10k.json 2> $O/bench_instanced.err
echo inst done
