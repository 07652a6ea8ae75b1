#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r3k
for v in 1 0 1 0; do
  MARIUS_FLASH_F16=$v timeout 300 python bench.py --no-cpu-baseline --no-fp32-pass --steps 300 > gpurun_out/r3k/bench_f16_$v.json 2> gpurun_out/r3k/bench_f16_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3k/bench_f16_$v.json"))
    print("f16=$v", d["ms_per_step"], d["loss_last_batch"], {k:(v["avg_ms"]) for k,v in d["kernels"].items() if k.startswith("lp_")})
except Exception as e:
    print("FAILED", e); print(open("gpurun_out/r3k/bench_f16_$v.err").read()[-2000:])
PY
done
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_flash.py 2>&1 | tail -2
