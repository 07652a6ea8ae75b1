#!/bin/bash
# A/B of environment settings on ONE box, interleaved with the default:  bash tools/ab_env.sh <tag> <reps> "NAME=VAL ..." "NAME2=VAL2" ...
tag=$1; reps=$2; shift 2
ulimit -c 0
out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; envs=$2; rep=$3
  env $envs timeout 300 python bench.py --steps 100 --no-arith-check --no-cpu-baseline --no-fp32-pass $BENCH_ARGS > $out/${name}_$rep.json 2> $out/${name}_$rep.err
  python -c "
import json; d=json.load(open('$out/${name}_$rep.json')); print('%-28s' % '$envs', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items() if k in ('lp_grad_adj','lp_grad_neg','lp_prep','lp_edge_bwd','segment_adagrad_scatter','flash_pack_neg')})" || tail -3 $out/${name}_$rep.err; }
for rep in $(seq 1 $reps); do
  run default "MARIUS_NOP=1" $rep
  i=0
  for v in "$@"; do i=$((i+1)); run v$i "$v" $rep; done
done
