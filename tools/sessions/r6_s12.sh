#!/bin/bash
# round 6 session 12: process-wide auxiliary streams: the driver's command with / without the check before it; host + sharded tests
tag=${1:-r6s12}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
run() { name=$1; shift; timeout 900 python bench.py --gpus 1 --no-cpu-baseline --no-fp32-pass "$@" > $out/$name.json 2> $out/$name.err; python -c "
import json
d=json.load(open('$out/$name.json')); print('$name', d['ms_per_step'], d['arith_check']['ok'] if d.get('arith_check') else None)" 2>&1 | tail -1; }
run a_driver --steps 20 --warmup 5
run b_nocheck --steps 20 --warmup 5 --no-arith-check
run c_driver --steps 20 --warmup 5
run d_nocheck_100 --steps 100 --warmup 10 --no-arith-check
MARIUS_MAPS=fused run e_driver_fused --steps 20 --warmup 5
MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/bench_sharded_w1.json 2> $out/bench_sharded_w1.err; python -c "
import json
d=json.load(open('$out/bench_sharded_w1.json')); print('sharded_w1', d['ms_per_step'])"
timeout 1200 python -m pytest tests/test_gpu_host.py tests/test_gpu_sharded2.py tests/test_gpu_fullshape.py -q -m gpu -p no:cacheprovider > $out/pytest_host.txt 2>&1; tail -3 $out/pytest_host.txt | cut -c1-300; grep -E "^(FAILED|ERROR)" $out/pytest_host.txt | cut -c1-300 | head
