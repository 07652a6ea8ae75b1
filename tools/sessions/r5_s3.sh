#!/bin/bash
# round 5, session 3: fixed-capacity exchange — sharded tests, forced-sharded world-1 bench (fixed vs exact), timeline of the fixed form
tag=${1:-s3}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_gpu_host.py tests/test_gpu_sharded2.py tests/test_gpu_parity.py -q -m gpu -x -k "sharded or segment or adagrad or planned or grouped or merge" -p no:cacheprovider > $out/pytest.txt 2>&1; tail -15 $out/pytest.txt | cut -c1-250
for mode in fixed exact; do
  MARIUS_EXCHANGE=$mode MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/bench_sharded_w1_$mode.json 2> $out/bench_sharded_w1_$mode.err
  python - <<PY
import json
try:
    d=json.load(open("$out/bench_sharded_w1_$mode.json")); print("sharded w1 $mode", d["ms_per_step"], d.get("host_issue_ms_per_step"), d.get("host_phase_ms_per_step"), d["dtype"][:40])
except Exception as e: print("sharded failed", e); print(open("$out/bench_sharded_w1_$mode.err").read()[-3000:])
PY
done
( cd /tmp && MARIUS_FORCE_SHARDED=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --no-arith-check --steps 40 --warmup 10 > $out/kt.log 2>&1 )
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f lp_prep2_kernel 30 > $out/timeline_sharded_w1.txt
python tools/trace_kernel_table.py $f lp_prep2_kernel > $out/kernel_table_sharded_w1.txt
head -45 $out/kernel_table_sharded_w1.txt | cut -c1-170
