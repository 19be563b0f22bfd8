#!/bin/bash
# Round-4 measurement set on one GPU box: PMC records (-> profiles/r04_traffic.json, tied to the library sources), kernel traces, bench lines.
# usage: bash scripts/r04_profiles.sh [A|B|C|ABC]   (from the repo root on the GPU box; everything lands in gpurun_out/r04p/ and is copied to profiles/ afterwards by the caller)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04p; mkdir -p $O
cd $R
pmc() {  # tag, waves per SIMD, bench args...
  T=$1; W=$2; shift 2
  bash scripts/pmc_profile.sh r04p_$T "$@" > $O/pmc_$T.log 2>&1
  cp gpurun_out/pmc_r04p_$T/summary.txt $O/r04_pmc_summary_$T.txt
  python3 scripts/traffic_merge.py gpurun_out/pmc_r04p_$T/traffic.json profiles/r04_pmc_summary_$T.txt $W >> $O/merge.log 2>&1
  echo "pmc $T done"
}
# stages (a gpurun call lasts 20 minutes at most): A = PMC of configs[2], [1], 1M; B = PMC of configs[4], [3], the instanced workload; C = kernel traces + bench lines
STAGE=${1:-ABC}
if [[ $STAGE == *A* ]]; then
pmc config2 6 --config 2
pmc config1 6 --config 1
pmc 1M 6 --config 1M
fi
if [[ $STAGE == *B* ]]; then
pmc config4 5 --config 4
pmc config3 6 --config 3
pmc instanced_1000x10k 5 --config 3 --instances 1000 --n-tris 10000
fi
cp profiles/r04_traffic.json $O/r04_traffic.json
if [[ $STAGE != *C* ]]; then exit 0; fi
bash scripts/kstats.sh r04p_k2 --config 2 --steps 3 --warmup 1 > $O/kstats_config2.txt 2>&1; cp $(ls gpurun_out/r04p_k2_stats/*/*kernel_stats.csv | tail -1) $O/r04_kernel_stats_config2.csv
bash scripts/kstats.sh r04p_k4 --config 4 --steps 1 --warmup 1 > $O/kstats_config4.txt 2>&1; cp $(ls gpurun_out/r04p_k4_stats/*/*kernel_stats.csv | tail -1) $O/r04_kernel_stats_config4.csv
echo "kernel traces done"
python3 bench.py > $O/r04_bench_config2.json 2> $O/bench_config2.err; echo "bench config2 done"
python3 bench.py --config 4 --steps 3 > $O/r04_bench_config4.json 2> $O/bench_config4.err; echo "bench config4 done"
python3 bench.py --config 3 > $O/r04_bench_config3_n1.json 2> $O/bench_config3.err
python3 bench.py --config 1M --steps 10 > $O/r04_bench_1M.json 2> $O/bench_1M.err
python3 bench.py --config 1 --steps 10 > $O/r04_bench_config1.json 2> $O/bench_config1.err
python3 bench.py --config 3 --instances 1000 --n-tris 10000 > $O/r04_bench_instanced_1000x10k.json 2> $O/bench_inst.err
python3 bench.py --gpus 2 --backend gloo --config 3 --steps 2 > $O/r04_bench_config3_2ranks_one_gpu_gloo.json 2> $O/bench_2r.err
python3 bench.py --gpus 2 --multi-handle --backend gloo --config 3 --steps 2 > $O/r04_bench_config3_multi_handle_2ctx_one_gpu.json 2> $O/bench_mh.err
echo "bench lines done"
for f in $O/r04_bench_*.json; do python3 -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1]); r=d.get('roofline',{})
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['stage_ms_per_step_rank0'], 'frac', r.get('frac'), 'valu', r.get('valu'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))" $f; done
