#!/bin/bash
# usage bash scripts/r02_refresh_profiles.sh <tag>: kernel trace of the default bench line + PMC passes of configs[2] / configs[3] (then traffic_merge.py on the host, then r02_bench_lines.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline-count > $O/stats_config2.json 2> $O/stats_config2.err && echo trace done
cd $R
bash scripts/pmc_profile.sh ${T}_config2 > $O/pmc_config2.log 2>&1 && echo pmc c2 done
bash scripts/pmc_profile.sh ${T}_config3 --config 3 > $O/pmc_config3.log 2>&1 && echo pmc c3 done
