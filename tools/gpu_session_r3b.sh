#!/bin/bash
# round-3 GPU session B: column-tail flash kernels — parity, A/B bench, kernel trace
ulimit -c 0
mkdir -p gpurun_out/r3b
echo "== flash parity"; timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_flash.py tests/test_gpu_fullshape.py -k "flash or cpp_trainer" 2>&1 | tail -15 | tee gpurun_out/r3b/tests_flash.txt
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_host.py -k "trainer_epoch or sharded" 2>&1 | tail -5 | tee -a gpurun_out/r3b/tests_flash.txt
for v in 1 0; do
  echo "== bench MARIUS_FLASH_TAIL=$v"
  MARIUS_FLASH_TAIL=$v timeout 300 python bench.py --no-cpu-baseline --no-fp32-pass --steps 200 > gpurun_out/r3b/bench_tail$v.json 2> gpurun_out/r3b/bench_tail$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3b/bench_tail$v.json"))
print(d["ms_per_step"], d["loss_last_batch"], {k:(v["avg_ms"]) for k,v in d["kernels"].items()})
PY
done
echo "== rocprofv3 kernel trace (default)"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_r3b -o r3b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-pass --steps 100 > /tmp/prof_r3b.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_r3b -name "*kernel_stats.csv" | head -1); echo $f; cp $f gpurun_out/r3b/kernel_stats.csv 2>/dev/null; head -25 gpurun_out/r3b/kernel_stats.csv | cut -c1-170
