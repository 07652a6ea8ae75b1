#!/bin/bash
# round 5, session 13: forced-sharded world 1 — where the MT19937 fills run (exchange stream pool / preparation stream inline / own stream)
tag=${1:-s13}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_gpu_host.py tests/test_gpu_sharded2.py -q -m gpu -x -k "sharded" -p no:cacheprovider > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt | cut -c1-300
run() { name=$1; shift
  env "$@" MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/b_$name.json 2> $out/b_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$out/b_$name.json")); print("%-16s" % "$name", d["ms_per_step"], "busy", d.get("host_busy_ms_per_step"), d.get("host_phase_ms_per_step"))
except Exception as e: print("$name failed", e); print(open("$out/b_$name.err").read()[-1500:])
PY
}
for rep in 1 2; do run xchg_pool_$rep A=1; run prep_inline_$rep MARIUS_MT_FILL=prep; done
run own_stream MARIUS_MT_FILL=own
( cd /tmp && MARIUS_FORCE_SHARDED=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --no-arith-check --steps 40 --warmup 10 > $out/kt.log 2>&1 )
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f lp_prep2_kernel 30 > $out/timeline.txt
python tools/trace_kernel_table.py $f lp_prep2_kernel > $out/kernel_table.txt
head -8 $out/kernel_table.txt
