#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r3o
for v in 1 0 1 0; do
MARIUS_LOADER_GATE=$v timeout 300 python bench.py --no-cpu-baseline --no-fp32-pass --steps 300 > gpurun_out/r3o/bench_gate$v.json 2> gpurun_out/r3o/bench_gate$v.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3o/bench_gate$v.json"))
    print("gate=$v", d["ms_per_step"], d["loss_last_batch"], {k:(v["avg_ms"]) for k,v in d["kernels"].items() if k.startswith("lp_grad") or k in ("sort_unique",)})
except Exception as e:
    print("FAILED", e); print(open("gpurun_out/r3o/bench_gate$v.err").read()[-2000:])
PY
done
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_host.py tests/test_gpu_fullshape.py tests/test_gpu_partition.py -k "trainer or cpp or partitioned_epochs or ten_million" 2>&1 | tail -3
