#!/bin/bash
# round-3 GPU session A: validate the f1 fix, the new tests, smoke, bench; cfg5-scale out-of-core run on HEAD
ulimit -c 0
mkdir -p gpurun_out/r3a
echo "== probe"; timeout 200 python tools/probe_torch_sort.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3a/probe_torch_sort.txt
echo "== tests"; timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_partition.py "tests/test_gpu_fullshape.py::test_cpp_trainer_flash_pipeline_bench_shape_matches_cpu_step" \
   tests/test_gpu_flash.py -k "partition or cpp_trainer or split_error_bound or deterministic" -s 2>&1 | grep -v "^$" | tail -40 | tee gpurun_out/r3a/tests.txt
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_parity.py -k "sort or unique or planned or merge" 2>&1 | tail -5 | tee -a gpurun_out/r3a/tests.txt
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench"; timeout 400 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; tail -c 3000 gpurun_out/r3a/bench.json; tail -3 gpurun_out/r3a/bench.err
echo "== sort micro"; timeout 100 python tools/bench_sort.py 2>&1 | tail -4
echo "== cfg5 scale d=400"
PB_TRACE=1 PB_TRACE_DUMP_AFTER=500 timeout 560 python tools/bench_partition_train.py --nodes 20000000 --d 400 --relations 1 --partitions 16 --capacity 8 --edges 260000000 --dir /dev/shm \
    > gpurun_out/r3a/partition_cfg5.json 2> gpurun_out/r3a/partition_cfg5.err
echo rc=$?; grep -v amdgpu.ids gpurun_out/r3a/partition_cfg5.err | tail -12; cat gpurun_out/r3a/partition_cfg5.json
rm -f /dev/shm/pb_bench_*
