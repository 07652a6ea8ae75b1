#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r3j
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_flash.py tests/test_gpu_parity.py -k "flash or tracked or planned" -s > gpurun_out/r3j/tests_flash.txt 2>&1; tail -3 gpurun_out/r3j/tests_flash.txt
grep -n "worst |err|" gpurun_out/r3j/tests_flash.txt | tail -10
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_host.py tests/test_gpu_fullshape.py -k "trainer_epoch or cpp_trainer or train_steps" -s > gpurun_out/r3j/tests_host.txt 2>&1; tail -3 gpurun_out/r3j/tests_host.txt
grep -n "touched\|relations \|loss of" gpurun_out/r3j/tests_host.txt | tail -12
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for v in 1 0; do
  MARIUS_FLASH_F16=$v timeout 300 python bench.py --no-cpu-baseline --no-fp32-pass --steps 200 > gpurun_out/r3j/bench_f16_$v.json 2> gpurun_out/r3j/bench_f16_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3j/bench_f16_$v.json"))
    print("f16=$v", d["ms_per_step"], d["loss_last_batch"], {k:(v["avg_ms"]) for k,v in d["kernels"].items()})
except Exception as e:
    print("FAILED", e); print(open("gpurun_out/r3j/bench_f16_$v.err").read()[-2000:])
PY
done
