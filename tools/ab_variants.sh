#!/bin/bash
# A/B of build_abl/<name>/libmarius_hip.so variants against the default build on ONE box, interleaved:  [BENCH_ARGS="..."] bash tools/ab_variants.sh <tag> <reps> name1 name2 ...
tag=$1; reps=$2; shift 2
ulimit -c 0
out=gpurun_out/$tag; mkdir -p $out
KEYS="('lp_scores','lp_grad_adj','lp_grad_neg','lp_prep','lp_pack','lp_edge_bwd','segment_adagrad_scatter','lp_lse')"
run() { name=$1; lib=$2; rep=$3
  if [ -n "$lib" ]; then export LD_LIBRARY_PATH=$PWD/build_abl/$lib MARIUS_HIP_LIB=$PWD/build_abl/$lib/libmarius_hip.so; else unset LD_LIBRARY_PATH MARIUS_HIP_LIB; fi
  timeout 300 python bench.py --steps 100 --no-arith-check --no-cpu-baseline --no-fp32-pass $BENCH_ARGS > $out/${name}_$rep.json 2> $out/${name}_$rep.err
  python -c "
import json; d=json.load(open('$out/${name}_$rep.json')); print('%-12s' % '$name', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items() if k in $KEYS})" || tail -3 $out/${name}_$rep.err; }
for rep in $(seq 1 $reps); do
  run default "" $rep
  for v in "$@"; do run $v $v $rep; done
done
