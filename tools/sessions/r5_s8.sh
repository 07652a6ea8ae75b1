#!/bin/bash
# round 5, session 8: both exchange forms through the sharded tests + the exchange kernel parity test; world-1 bench of both forms
tag=${1:-s8}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1200 python -m pytest tests/test_gpu_host.py tests/test_gpu_sharded2.py tests/test_gpu_parity.py -q -m gpu -k "sharded or fixed_capacity or merge or planned or grouped" -p no:cacheprovider > $out/pytest.txt 2>&1; tail -12 $out/pytest.txt | cut -c1-300
run() { name=$1; shift
  env "$@" MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/b_$name.json 2> $out/b_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$out/b_$name.json")); print("%-22s" % "$name", d["ms_per_step"], "issue", d.get("host_issue_ms_per_step"), "busy", d.get("host_busy_ms_per_step"), d.get("host_phase_ms_per_step"))
except Exception as e: print("$name failed", e); print(open("$out/b_$name.err").read()[-1500:])
PY
}
run exact A=1
run fixed MARIUS_EXCHANGE=fixed
run exact_free0 MARIUS_SHARDED_FREE_CUS=0
run exact_free48 MARIUS_SHARDED_FREE_CUS=48
run exact_b A=1
