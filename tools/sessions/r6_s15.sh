#!/bin/bash
# round 6 session 15: whole GPU suite; the driver's command (check deferred behind the timed region) twice
tag=${1:-r6s15}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
for i in 1 2; do
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_$i.json 2> $out/bench_driver_$i.err
python - <<PY
import json
try:
    d=json.load(open("$out/bench_driver_$i.json")); a=d["arith_check"]; print("driver cmd", d["ms_per_step"], "arith ok", a["ok"], a.get("evaluated"), "seconds", a["seconds"], "fp32", d["fp32_exact"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
except Exception as e: print("bench failed", e); print(open("$out/bench_driver_$i.err").read()[-3000:])
PY
done
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt | cut -c1-300
grep -E "^(FAILED|ERROR)" $out/pytest_gpu.txt | cut -c1-300 | head -60
