#!/bin/bash
# A/B of environment settings on one box:  bash scripts/ab_env.sh TAG "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...   ("-" = no setting)
# prints value, traversal / shade ms per frame and the film hash per setting; raw lines in gpurun_out/TAG_ab.jsonl
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; ARGS=$2; shift 2
: > $R/gpurun_out/${TAG}_ab.jsonl
for SET in "$@"; do
  if [ "$SET" = "-" ]; then SETV=""; else SETV="$SET"; fi
  OUT=$(env $SETV python3 $R/bench.py --no-cpu-baseline --no-roofline-count $ARGS 2>>$R/gpurun_out/${TAG}_ab.err | tail -1)
  echo "{\"env\": \"$SET\", \"line\": $OUT}" >> $R/gpurun_out/${TAG}_ab.jsonl
  python3 -c "
import json,sys
d=json.loads(sys.argv[1]); s=d['stage_ms_per_step_rank0']
print('%-44s %9.1f Mrays/s  trav %9.2f ms  rest %8.2f ms  %s' % (sys.argv[2], d['value'], s['traversal'], s['raygen_shade_film'], d['film_sha256'][:12]))" "$OUT" "$SET"
done
