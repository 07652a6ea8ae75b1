#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r3m
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3m/pytest_gpu.txt 2>&1; grep -E "passed|failed" gpurun_out/r3m/pytest_gpu.txt | tail -2; grep -n "^FAILED\|^E  " gpurun_out/r3m/pytest_gpu.txt | head -8
timeout 400 python bench.py --workload twitter --steps 50 > gpurun_out/r3m/bench_twitter.json 2> gpurun_out/r3m/bench_twitter.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3m/bench_twitter.json"))
    print("twitter", d["ms_per_step"], d["positive_edges_per_s"], {k:(v["avg_ms"]) for k,v in d["kernels"].items()})
except Exception as e:
    print("FAILED", e); print(open("gpurun_out/r3m/bench_twitter.err").read()[-1500:])
PY
