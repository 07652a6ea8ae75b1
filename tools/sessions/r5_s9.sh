#!/bin/bash
# round 5, session 9: flash tests under the tiered mixed_close + the driver's bench command (one lease of the README's range)
tag=${1:-s9}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_gpu_flash.py -q -m gpu -p no:cacheprovider > $out/pytest_flash.txt 2>&1; tail -5 $out/pytest_flash.txt | cut -c1-300
grep -E "^(FAILED|ERROR)" $out/pytest_flash.txt | cut -c1-250 | head -30
grep -E "AssertionError: " $out/pytest_flash.txt | sort | uniq -c | sort -rn | head -20 | cut -c1-200
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
python - <<PY
import json
try:
    d=json.load(open("$out/bench_driver.json")); print("driver cmd", d["ms_per_step"], "arith ok", d["arith_check"]["ok"], {k:v for k,v in d["arith_check"]["verdict"].items() if k!="rule"}, "fp32_exact", d["fp32_exact"]["ms_per_step"])
except Exception as e: print("bench failed", e); print(open("$out/bench_driver.err").read()[-2000:])
PY
