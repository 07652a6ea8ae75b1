#!/bin/bash
# round 5, session 2: does un-sharing the HSA queues (GPU_MAX_HW_QUEUES) fix the forced-sharded step?   usage: bash tools/sessions/r5_s2.sh <tag>
tag=${1:-s2}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/bench_sharded_w1_q$q.json 2> $out/bench_sharded_w1_q$q.err
  python - <<PY
import json
try:
    d=json.load(open("$out/bench_sharded_w1_q$q.json")); print("sharded w1 queues=$q", d["ms_per_step"], d.get("host_issue_ms_per_step"), d.get("host_phase_ms_per_step"))
except Exception as e: print("sharded failed", e); print(open("$out/bench_sharded_w1_q$q.err").read()[-2000:])
PY
done
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-arith-check --no-fp32-pass > $out/bench_q8.json 2> $out/bench_q8.err
python -c "
import json; d=json.load(open('$out/bench_q8.json')); print('fused q8', d['ms_per_step'])"
( cd /tmp && GPU_MAX_HW_QUEUES=8 MARIUS_FORCE_SHARDED=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --no-arith-check --steps 40 --warmup 10 > $out/kt.log 2>&1 )
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f lp_prep2_kernel 30 > $out/timeline_sharded_w1_q8.txt
python tools/trace_kernel_table.py $f lp_prep2_kernel > $out/kernel_table_sharded_w1_q8.txt
head -12 $out/kernel_table_sharded_w1_q8.txt | cut -c1-170
