#!/bin/bash
# usage bash scripts/r02_fuzz_sweep.sh <tag> <first> <last> <tex_first> <tex_last>: the two fuzz generators in chunks (a line of progress per chunk)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
for ((a=$2; a<$3; a+=1000)); do
  b=$((a+1000)); [ $b -gt $3 ] && b=$3
  timeout -k 10 420 python3 scripts/fuzz_many.py $a $b >> $O/fuzz.log 2>&1 || { echo "fuzz $a $b FAILED"; tail -3 $O/fuzz.log; exit 1; }
  echo "fuzz $a..$b: $(tail -1 $O/fuzz.log)"
done
for ((a=$4; a<$5; a+=400)); do
  b=$((a+400)); [ $b -gt $5 ] && b=$5
  timeout -k 10 420 python3 scripts/fuzz_textures_many.py $a $b >> $O/fuzz_tex.log 2>&1 || { echo "fuzz_tex $a $b FAILED"; tail -3 $O/fuzz_tex.log; exit 1; }
  echo "fuzz_tex $a..$b: $(tail -1 $O/fuzz_tex.log)"
done
