#!/bin/bash
# Second PMC set (instruction mix, issue side, vector-memory queueing, TCP busy and stall cycles; the TA_* counters are left out: a pass with them never returned on this pool, gpurun r03r) over one short bench run per pass — separate runs per counter group, no trace domains.
# usage: scripts/pmc_extra.sh <tag> [bench args...]   -> gpurun_out/pmcx_<tag>/summary.txt
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmcx_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline-count $@"
i=0
for SET in \
  "SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM" \
  "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $R/bench.py $ARGS > $OUT/pass$i.json 2> $OUT/pass$i.err || { tail -5 $OUT/pass$i.err; echo "pass $i failed"; }
done
python3 $R/scripts/pmc_aggregate.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
