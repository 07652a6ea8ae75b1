#!/bin/bash
# per-stream timeline of one steady-state step of the default bench command (rocprofv3 kernel trace -> tools/trace_gaps.py)
# usage (GPU box): bash tools/sessions/gpu_session_timeline.sh <tag> [bench args]   -> gpurun_out/<tag>/timeline.txt
tag=${1:-tl}; shift
ulimit -c 0
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-pass --no-arith-check --no-profile --steps 40 --warmup 10 "$@" > $out/kt.log 2>&1 )
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_gaps.py $f lp_prep2 30 > $out/timeline.txt
cat $out/timeline.txt | cut -c1-150
