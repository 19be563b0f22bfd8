#!/bin/bash
# Instruction-class mix per kernel (one short bench run per pass):  bash scripts/pmc_mix.sh TAG <bench args>  -> gpurun_out/pmcm_TAG/summary.txt
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmcm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline-count $@"
i=0
for SET in \
  "SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH" ; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $R/bench.py $ARGS > $OUT/pass$i.json 2> $OUT/pass$i.err || { tail -5 $OUT/pass$i.err; echo "pass $i failed"; }
done
python3 $R/scripts/pmc_aggregate.py $OUT > $OUT/summary.txt
