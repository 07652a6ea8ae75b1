#!/bin/bash
# round 5 evidence of the committed tree: default workload (bench line, rocprofv3 kernel stats, PMC traffic, one-step timeline), the twitter
# (cfg5 shape) and degree-fraction-0.5 workloads (bench line, kernel stats, PMC traffic), forced-sharded world 1 (bench line, timeline, kernel
# table), and the driver's own command twice.   usage (GPU box): bash tools/sessions/evidence.sh <tag>
tag=${1:-r5ev}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
pmc() {  # name, bench args...
  name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${tag}_${name}_$c -o pmc --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-pass --no-arith-check --no-profile "$@" > $out/pmc_${name}_$c.log 2>&1 )
    f=$(find /tmp/pmc_${tag}_${name}_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc_${name}_$c.csv
  done
  python tools/pmc_traffic.py $out/pmc_${name}_FETCH_SIZE.csv $out/pmc_${name}_WRITE_SIZE.csv $out/pmc_traffic_$name.json && rm -f $out/pmc_${name}_FETCH_SIZE.csv $out/pmc_${name}_WRITE_SIZE.csv
}
stats() {  # name, bench args...
  name=$1; shift
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_$name -o kt -- python $R/bench.py --no-cpu-baseline --no-fp32-pass --no-arith-check --steps 100 "$@" > $out/kt_$name.log 2>&1 )
  db=$(find /tmp/prof_${tag}_$name -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db > $out/kernel_stats_$name.txt && head -14 $out/kernel_stats_$name.txt | cut -c1-150
}
# ---- default workload
timeout 600 python bench.py > $out/bench_freebase86m.json 2> $out/bench_freebase86m.err; tail -c 300 $out/bench_freebase86m.json; echo
stats freebase86m
pmc freebase86m
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/tl_${tag} -o kt --output-format csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-pass --no-profile --no-arith-check > $out/kt_timeline.log 2>&1 )
f=$(find /tmp/tl_${tag} -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $f lp_prep2 -12 > $out/timeline_one_step.txt; head -3 $out/timeline_one_step.txt
# ---- twitter (cfg5 shape)
timeout 400 python bench.py --workload twitter --no-arith-check --no-cpu-baseline > $out/bench_twitter_d400.json 2> $out/bench_twitter_d400.err
stats twitter_d400 --workload twitter
pmc twitter --workload twitter
# ---- degree_fraction 0.5
timeout 400 python bench.py --steps 100 --degree-fraction 0.5 --no-arith-check --no-cpu-baseline --no-fp32-pass > $out/bench_freebase86m_deg05.json 2> $out/bench_deg05.err
pmc deg05 --degree-fraction 0.5
# ---- forced-sharded world 1
MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/bench_sharded_w1.json 2> $out/bench_sharded_w1.err
( cd /tmp && MARIUS_FORCE_SHARDED=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_${tag}_sh -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --no-arith-check --steps 40 --warmup 10 > $out/kt_sharded.log 2>&1 )
f=$(find /tmp/tl_${tag}_sh -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f lp_prep2 30 > $out/sharded_timeline.txt
python tools/trace_kernel_table.py $f lp_prep2 > $out/sharded_kernel_table.txt
# ---- the driver's command, twice more
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_$i.json 2> $out/bench_driver_$i.err; done
python - <<PY
import json
for f in ("bench_freebase86m","bench_twitter_d400","bench_freebase86m_deg05","bench_sharded_w1","bench_driver_1","bench_driver_2"):
    try:
        d=json.load(open("$out/%s.json"%f)); print(f, d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
    except Exception as e: print(f, "FAILED", e)
PY
