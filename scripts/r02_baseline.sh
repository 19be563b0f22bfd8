#!/bin/bash
# Round-2 baseline on the GPU box: parity tests, the bench lines of configs[2] / configs[3] / 1 M / configs[1], a kernel trace and PMC passes.
# usage (on the GPU box, from the repo root): bash scripts/r02_baseline.sh <tag>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r02a}
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python bench.py --steps 5 --warmup 1 > $O/bench_config2.json 2> $O/bench_config2.err; echo c2 done
python bench.py --config 3 --steps 3 --warmup 1 > $O/bench_config3_n1.json 2> $O/bench_config3.err; echo c3 done
python bench.py --config 1M --steps 5 --warmup 2 > $O/bench_1M.json 2> $O/bench_1M.err; echo 1M done
python bench.py --config 1 --steps 10 --warmup 2 > $O/bench_config1.json 2> $O/bench_config1.err; echo c1 done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline-count > $O/stats_config2.json 2> $O/stats_config2.err; echo trace done
cd $R
bash scripts/pmc_profile.sh ${T}_config2 > $O/pmc_config2.log 2>&1; echo pmc c2 done
bash scripts/pmc_profile.sh ${T}_config3 --config 3 > $O/pmc_config3.log 2>&1; echo pmc c3 done
