#!/bin/bash
# usage bash scripts/r02_bench_lines.sh <tag>: the four bench lines of profiles/r02_bench_*.json (bench.py reads the traffic entries of profiles/r02_traffic.json)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
python3 bench.py --steps 5 --warmup 1 > $O/bench_config2.json 2> $O/bench_config2.err && echo c2 done
python3 bench.py --config 3 --steps 3 --warmup 1 > $O/bench_config3_n1.json 2> $O/bench_config3.err && echo c3 done
python3 bench.py --config 1M --steps 5 --warmup 2 > $O/bench_1M.json 2> $O/bench_1M.err && echo 1M done
python3 bench.py --config 1 --steps 10 --warmup 2 > $O/bench_config1.json 2> $O/bench_config1.err && echo c1 done
