#!/bin/bash
# usage bash scripts/r02_ab7.sh <tag> [configs...]: the parity tests that cover the shade kernel, then the default bench lines twice (shade-kernel A/B across builds)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_render_gpu.py tests/test_materials_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for rep in 1 2; do for CFG in "${@:-2 1}"; do
  python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-count > $O/c${CFG}_$rep.json 2> $O/c${CFG}_$rep.err || { echo "config $CFG FAILED"; tail -3 $O/c${CFG}_$rep.err; continue; }
  python3 - <<PY
import json
d=json.loads([l for l in open('$O/c${CFG}_$rep.json').read().splitlines() if l.startswith('{')][-1])
print('config $CFG', d['value'], d['ms_per_step'], d['stage_ms_per_step_rank0'], d['film_sha256'][:12])
PY
done; done
