#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r3e
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_flash.py -s > gpurun_out/r3e/tests_flash.txt 2>&1; tail -4 gpurun_out/r3e/tests_flash.txt
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_host.py tests/test_gpu_parity.py tests/test_gpu_fullshape.py -k "trainer_epoch or sharded or train_steps or cpp_trainer or sort or unique or merge" 2>&1 | tail -5 | tee gpurun_out/r3e/tests_other.txt
echo "== bench f=0 default"; timeout 300 python bench.py --no-cpu-baseline --no-fp32-pass --steps 200 > gpurun_out/r3e/bench.json 2>gpurun_out/r3e/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r3e/bench.json"))
print(d["ms_per_step"], d["loss_last_batch"], {k:(v["avg_ms"]) for k,v in d["kernels"].items()})
PY
timeout 100 python tools/bench_sort.py 2>&1 | tail -2
