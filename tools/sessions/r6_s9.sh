#!/bin/bash
# round 6 session 9: the driver's 20-step window after the arithmetic check's pretraining: steady or one-time? allocator?
tag=${1:-r6s9}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
run() { name=$1; shift; timeout 900 python bench.py --gpus 1 --no-cpu-baseline --no-fp32-pass --step-groups 8 "$@" > $out/$name.json 2> $out/$name.err; python -c "
import json
d=json.load(open('$out/$name.json')); print('$name', d['ms_per_step'], 'allocs in timed region', d['device_allocations_in_timed_region'], [ (g['ms_per_step'], g['device_allocs']) for g in d['step_groups']])" 2>&1 | tail -1; }
run a_driver --steps 20 --warmup 5
run b_nocheck --steps 20 --warmup 5 --no-arith-check
MARIUS_BENCH_EMPTY_CACHE=1 run c_driver_empty --steps 20 --warmup 5
