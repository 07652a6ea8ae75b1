#!/bin/bash
# round 6 session 8: where do the 60 us per step of the driver's 20-step command go? (with / without the arithmetic check's pretraining, short / long warm-up)
tag=${1:-r6s8}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
run() { name=$1; shift; timeout 900 python bench.py --gpus 1 --no-cpu-baseline --no-fp32-pass "$@" > $out/$name.json 2> $out/$name.err; python -c "
import json
d=json.load(open('$out/$name.json')); print('$name', d['ms_per_step'], d.get('host_issue_ms_per_step'), d.get('loss_last_batch'))" 2>&1 | tail -1; }
run a_driver --steps 20 --warmup 5
run b_nocheck --steps 20 --warmup 5 --no-arith-check
run c_driver_w30 --steps 20 --warmup 30
run d_nocheck_w30 --steps 20 --warmup 30 --no-arith-check
run e_driver_again --steps 20 --warmup 5
run f_nocheck_100 --steps 100 --warmup 10 --no-arith-check
