#!/bin/bash
tag=${1:-r3h}
ulimit -c 0
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
( cd /tmp && MARIUS_FORCE_SHARDED=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 > $out/kt.log 2>&1 )
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db > $out/kernel_stats_sharded_w1.txt && head -40 $out/kernel_stats_sharded_w1.txt | cut -c1-150
tail -2 $out/kt.log | cut -c1-400
