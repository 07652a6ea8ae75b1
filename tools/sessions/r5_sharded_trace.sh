#!/bin/bash
# round 5, session 1: baseline of this box with the driver's command, per-stream timeline + kernel table of the forced-sharded world-1 step,
# K8-vs-K16 fp16 MFMA rate.   usage (GPU box): bash tools/sessions/r5_s1.sh <tag>
tag=${1:-s1}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 60 tools/micro/mfma_k8_tail > $out/mfma_k8_tail.txt 2>&1; cat $out/mfma_k8_tail.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
python - <<PY
import json
try:
    d=json.load(open("$out/bench_driver.json")); print("driver cmd", d["ms_per_step"], d["arith_check"]["ok"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
except Exception as e: print("bench failed", e); print(open("$out/bench_driver.err").read()[-2000:])
PY
MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/bench_sharded_w1.json 2> $out/bench_sharded_w1.err
python - <<PY
import json
try:
    d=json.load(open("$out/bench_sharded_w1.json")); print("sharded w1", d["ms_per_step"], d.get("host_issue_ms_per_step"), d.get("host_phase_ms_per_step"))
except Exception as e: print("sharded failed", e); print(open("$out/bench_sharded_w1.err").read()[-2000:])
PY
( cd /tmp && MARIUS_FORCE_SHARDED=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --no-arith-check --steps 40 --warmup 10 > $out/kt.log 2>&1 )
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f lp_prep2 30 > $out/timeline_sharded_w1.txt
python tools/trace_kernel_table.py $f lp_prep2 > $out/kernel_table_sharded_w1.txt
head -60 $out/kernel_table_sharded_w1.txt | cut -c1-170
