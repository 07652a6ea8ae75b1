#!/bin/bash
ulimit -c 0
out=gpurun_out/r4s1; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 100 --no-arith-check --no-cpu-baseline --no-fp32-pass > $out/$name.json 2> $out/$name.err; python -c "
import json; d=json.load(open('$out/$name.json')); print('$name', d['ms_per_step'], d['host_issue_ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items() if k in ('lp_grad_adj','lp_grad_neg','sort_unique')})" || tail -3 $out/$name.err; }
run base X=1
run gate0 MARIUS_LOADER_GATE=0
run loaderthread0 MARIUS_LOADER_THREAD=0
run base2 X=1
run nwg480_res16 MARIUS_FLASH_NWG=480 
run rotate0 MARIUS_FLASH_ROTATE=0
