#!/bin/bash
# round 5, session 11: reduce-only job in the grouped update — parity, sharded tests, forced-sharded world-1 bench (A/B with MARIUS_REL_GROUP=0)
tag=${1:-s11}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py tests/test_gpu_sharded2.py -q -m gpu -k "group or sharded or segment or planned or fixed_capacity" -p no:cacheprovider > $out/pytest.txt 2>&1; tail -6 $out/pytest.txt | cut -c1-300
grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-250 | head
run() { name=$1; shift
  env "$@" MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/b_$name.json 2> $out/b_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$out/b_$name.json")); print("%-16s" % "$name", d["ms_per_step"], "busy", d.get("host_busy_ms_per_step"))
except Exception as e: print("$name failed", e); print(open("$out/b_$name.err").read()[-1500:])
PY
}
for rep in 1 2; do run group3_$rep A=1; run rel4_$rep MARIUS_REL_GROUP=0; done
