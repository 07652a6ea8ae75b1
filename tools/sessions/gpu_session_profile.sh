#!/bin/bash
# rocprofv3 evidence of the default bench command: kernel trace + stats, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes.
# usage (GPU box): bash tools/sessions/gpu_session_profile.sh <tag>     -> gpurun_out/<tag>/{bench.json, kernel_stats.txt, pmc_traffic.json}
tag=${1:-r3}
ulimit -c 0
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 400 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json; echo
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-pass --no-arith-check --steps 100 > $out/kt.log 2>&1 )
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db > $out/kernel_stats.txt && head -24 $out/kernel_stats.txt | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${tag}_$c -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-pass --no-arith-check --no-profile > $out/pmc_$c.log 2>&1 )
  f=$(find /tmp/pmc_${tag}_$c -name "*counter_collection.csv" | head -1); echo "$c -> $f"; [ -n "$f" ] && cp $f $out/pmc_$c.csv
done
python tools/pmc_traffic.py $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv $out/pmc_traffic.json && rm -f $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv
python - <<PY
import json
d=json.load(open("$out/pmc_traffic.json"))["kernels"]
tot=0
for k,v in sorted(d.items(), key=lambda kv:-kv[1]["hbm_bytes"])[:16]:
    print("%-60s %8.1f MB" % (k[:60], v["hbm_bytes"]/1e6))
PY
