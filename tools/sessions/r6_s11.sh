#!/bin/bash
# round 6 session 11: is the driver window slower after the arithmetic check because the device is hot (power / clocks)?
tag=${1:-r6s11}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
run() { name=$1; shift; timeout 900 python bench.py --gpus 1 --no-cpu-baseline --no-fp32-pass "$@" > $out/$name.json 2> $out/$name.err; python -c "
import json
d=json.load(open('$out/$name.json')); print('$name', d['ms_per_step'], d['arith_check']['seconds'] if d.get('arith_check') else None)" 2>&1 | tail -1; }
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6
run a_driver --steps 20 --warmup 5
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6
MARIUS_BENCH_COOL_S=20 run b_driver_cool20 --steps 20 --warmup 5
MARIUS_BENCH_COOL_S=3 run c_driver_cool3 --steps 20 --warmup 5
run d_driver_steps100 --steps 100 --warmup 5
