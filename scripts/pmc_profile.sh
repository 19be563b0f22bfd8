#!/bin/bash
# PMC passes over one short bench run each (separate runs per counter group, no trace domains mixed in).
# usage: scripts/pmc_profile.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/passN/...
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline-count $@"
i=0
for SET in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" \
  "FETCH_SIZE TCC_HIT_sum" \
  "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $R/bench.py $ARGS > $OUT/pass$i.json 2> $OUT/pass$i.err || { tail -5 $OUT/pass$i.err; echo "pass $i failed"; }
done
python3 $R/scripts/pmc_aggregate.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
