#!/bin/bash
# round 6 session 2: whole suite (all failures listed, tiers everywhere), driver bench with the multi-input arithmetic gate
tag=${1:-r6s2}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider > $out/pytest_gpu.txt 2>&1; tail -6 $out/pytest_gpu.txt | cut -c1-300
grep -E "^(FAILED|ERROR)" $out/pytest_gpu.txt | cut -c1-300 | head -60
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
python - <<PY
import json
try:
    d=json.load(open("$out/bench_driver.json")); a=d["arith_check"]; print("driver cmd", d["ms_per_step"], "arith ok", a["ok"], "seconds", a["seconds"]); print(json.dumps(a["verdict"]))
    for p in a["per_input"]: print(p["input"][:60], p["ok"], p["worst_rms_vs_reference"], p["worst_max_vs_reference"], p["le1_cpu_aten"], p["le1_device_aten"], {q: (v["ratio_rms"], v["ratio_max"]) for q, v in p["flash_vs_either_reference_evaluation"].items()})
except Exception as e: print("bench failed", e); print(open("$out/bench_driver.err").read()[-3000:])
PY
