#!/bin/bash
# usage: tools/abl_scores.sh <variant a|p|b> <ablate masks...>   (GPU box)
v=$1; shift
for ab in "$@"; do
  MARIUS_ABLATE=$ab MARIUS_SCORES=$v timeout 90 python bench.py --no-cpu-baseline --driver py --steps 10 --warmup 3 2>&1 | grep "^{" > /tmp/abl.json
  python - "$ab" <<'PY'
import json,sys
j=json.load(open('/tmp/abl.json')); print("ablate", sys.argv[1], "step", j["ms_per_step"], "scores", j["kernels"]["lp_scores"]["avg_ms"])
PY
done
