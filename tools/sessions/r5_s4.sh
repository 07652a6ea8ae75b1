#!/bin/bash
# round 5, session 4: forced-sharded world 1 — where does the MT19937 pool fill belong, relation step grouped or not, HSA queue count
tag=${1:-s4}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
run() { name=$1; shift
  env "$@" MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/b_$name.json 2> $out/b_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$out/b_$name.json")); print("%-28s" % "$name", d["ms_per_step"], d.get("host_issue_ms_per_step"), d.get("host_phase_ms_per_step"))
except Exception as e: print("$name failed", e); print(open("$out/b_$name.err").read()[-1500:])
PY
}
run default MARIUS_SHARDED_FINE=1
run mt_inline MARIUS_MT_PREFETCH=0 MARIUS_SHARDED_FINE=1
run mt_inline_rel4 MARIUS_MT_PREFETCH=0 MARIUS_REL_GROUP=0
run rel4 MARIUS_REL_GROUP=0
run queues5 GPU_MAX_HW_QUEUES=5
run queues5_mt_inline GPU_MAX_HW_QUEUES=5 MARIUS_MT_PREFETCH=0
run stale2 MARIUS_SHARDED_STALENESS=2
run default_again A=1
( cd /tmp && MARIUS_MT_PREFETCH=0 MARIUS_FORCE_SHARDED=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --no-arith-check --steps 40 --warmup 10 > $out/kt.log 2>&1 )
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f lp_prep2_kernel 30 > $out/timeline_mt_inline.txt
python tools/trace_kernel_table.py $f lp_prep2_kernel > $out/kernel_table_mt_inline.txt
head -30 $out/kernel_table_mt_inline.txt | cut -c1-170
