#!/bin/bash
# round 6 session 16: stream-ordering events without the system-scope fence: A/B in one lease, then the host tests
tag=${1:-r6s16}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
run() { name=$1; shift; timeout 900 python bench.py --gpus 1 --no-cpu-baseline --no-fp32-pass --no-arith-check "$@" > $out/$name.json 2> $out/$name.err; python -c "
import json
d=json.load(open('$out/$name.json')); print('$name', d['ms_per_step'], d['roofline']['avg_ms'])" 2>&1 | tail -1; }
for i in 1 2; do
MARIUS_EVENT_FENCE=system run sys20_$i --steps 20 --warmup 5
run dev20_$i --steps 20 --warmup 5
MARIUS_EVENT_FENCE=system run sys100_$i --steps 100 --warmup 10
run dev100_$i --steps 100 --warmup 10
done
MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/bench_sharded_w1.json 2> $out/bench_sharded_w1.err; python -c "
import json
d=json.load(open('$out/bench_sharded_w1.json')); print('sharded_w1', d['ms_per_step'])"
timeout 1200 python -m pytest tests/test_gpu_host.py tests/test_gpu_fullshape.py tests/test_gpu_sharded2.py tests/test_gpu_partition.py -q -m gpu -p no:cacheprovider > $out/pytest_host.txt 2>&1; tail -2 $out/pytest_host.txt | cut -c1-300; grep -E "^(FAILED|ERROR)" $out/pytest_host.txt | cut -c1-300 | head
