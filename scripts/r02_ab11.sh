#!/bin/bash
# usage bash scripts/r02_ab11.sh <tag> [bench args...]: two builds of the library (pbrt-v3-rs_amd/ab_old.so, ab_new.so) alternated on one bench line, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
for i in 1 2 3; do for v in ${VARS:-old new}; do
  cp pbrt-v3-rs_amd/ab_$v.so pbrt-v3-rs_amd/libpbrt_hip.so
  python3 bench.py "$@" --no-cpu-baseline --no-roofline-count > $O/$v$i.json 2> $O/$v$i.err || { echo "$v$i FAILED"; tail -3 $O/$v$i.err; continue; }
  python3 - <<PY
import json
d=json.loads([l for l in open('$O/$v$i.json').read().splitlines() if l.startswith('{')][-1])
print('$v $i', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['stage_ms_per_step_rank0'], d['film_sha256'][:12])
PY
done; done
cp pbrt-v3-rs_amd/ab_${LAST:-new}.so pbrt-v3-rs_amd/libpbrt_hip.so
