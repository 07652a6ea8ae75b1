#!/bin/bash
# round 6 session 13: one step of the timed region, kernel by kernel and queue by queue, with and without the arithmetic check before it
tag=${1:-r6s13}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
for v in check nocheck; do
  extra=""; [ $v = nocheck ] && extra="--no-arith-check"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/tl_${tag}_$v -o kt --output-format csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-pass --no-profile $extra > $out/kt_$v.log 2>&1 )
  f=$(find /tmp/tl_${tag}_$v -name "*kernel_trace.csv" | head -1)
  python tools/trace_gaps.py $f lp_prep2 -12 > $out/timeline_$v.txt
  python tools/trace_gaps.py $f lp_prep2 -8 > $out/timeline_${v}_b.txt
  grep ms_per_step $out/kt_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])"
  head -1 $out/timeline_$v.txt
done
