#!/bin/bash
# whole GPU suite (all failures listed) + smoke + the driver's bench command + forced-sharded world 1
tag=${1:-full}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $out/pytest_gpu.txt 2>&1; tail -6 $out/pytest_gpu.txt | cut -c1-300
grep -E "^(FAILED|ERROR)" $out/pytest_gpu.txt | cut -c1-250 | head -40
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
python - <<PY
import json
try:
    d=json.load(open("$out/bench_driver.json")); print("driver cmd", d["ms_per_step"], "arith ok", d["arith_check"]["ok"], d["arith_check"]["verdict"])
except Exception as e: print("bench failed", e); print(open("$out/bench_driver.err").read()[-2000:])
PY
MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/bench_sharded_w1.json 2> $out/bench_sharded_w1.err
python - <<PY
import json
try:
    d=json.load(open("$out/bench_sharded_w1.json")); print("sharded w1", d["ms_per_step"], "issue", d.get("host_issue_ms_per_step"), "busy", d.get("host_busy_ms_per_step"), d.get("host_phase_ms_per_step"))
except Exception as e: print("sharded failed", e); print(open("$out/bench_sharded_w1.err").read()[-2000:])
PY
