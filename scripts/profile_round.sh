set -e
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/final2_bench.json 2> gpurun_out/final2_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final2_stats -- python3 $R/bench.py --no-cpu-baseline --no-roofline-count > $R/gpurun_out/final2_stats.json 2> $R/gpurun_out/final2_stats.err
cd $R
bash scripts/pmc_profile.sh final2 > gpurun_out/pmc_final2.log 2>&1
