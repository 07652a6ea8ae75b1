#!/bin/bash
# round-3 GPU session C: fused forward + dAdj sweep — parity, A/B bench, kernel trace
ulimit -c 0
mkdir -p gpurun_out/r3c
echo "== flash parity"; timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_flash.py tests/test_gpu_fullshape.py -k "flash or cpp_trainer" -s > gpurun_out/r3c/tests_flash.txt 2>&1; tail -4 gpurun_out/r3c/tests_flash.txt
grep -n "largest (max\|touched table rows\|worst |err|" gpurun_out/r3c/tests_flash.txt | tail -12
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_host.py tests/test_gpu_parity.py -k "trainer_epoch or sharded or train_steps" 2>&1 | tail -3 | tee -a gpurun_out/r3c/tests_flash.txt
for v in 1 0; do
  echo "== bench MARIUS_FLASH_FUSED=$v"
  MARIUS_FLASH_FUSED=$v timeout 300 python bench.py --no-cpu-baseline --no-fp32-pass --steps 200 > gpurun_out/r3c/bench_fused$v.json 2> gpurun_out/r3c/bench_fused$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c/bench_fused$v.json"))
print(d["ms_per_step"], d["loss_last_batch"], {k:(v["avg_ms"]) for k,v in d["kernels"].items()})
PY
done
echo "== rocprofv3 kernel trace (default)"
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_r3c -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-pass --steps 100 > /tmp/prof_r3c.log 2>&1 )
find /tmp/prof_r3c -name "*stats*" | head; f=$(find /tmp/prof_r3c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r3c/kernel_stats.csv && head -22 gpurun_out/r3c/kernel_stats.csv | cut -c1-150
tail -3 /tmp/prof_r3c.log
