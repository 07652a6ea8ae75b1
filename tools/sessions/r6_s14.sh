#!/bin/bash
# round 6 session 14: the driver's window with / without the check, with / without the HIP-event bracket around the dominant kernel
tag=${1:-r6s14}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
run() { name=$1; shift; timeout 900 python bench.py --gpus 1 --no-cpu-baseline --no-fp32-pass --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err; python -c "
import json
d=json.load(open('$out/$name.json')); print('$name', d['ms_per_step'])" 2>&1 | tail -1; }
run a_check
run b_check_noprofile --no-profile
run c_nocheck --no-arith-check
run d_nocheck_noprofile --no-arith-check --no-profile
run e_check
run f_check_noprofile --no-profile
