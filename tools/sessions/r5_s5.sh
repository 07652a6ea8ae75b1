#!/bin/bash
# round 5, session 5: where the host time of ShardedTrainer::prepare goes
tag=${1:-s5}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
run() { name=$1; shift
  env "$@" MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/b_$name.json 2> $out/b_$name.err
  grep "sharded fine" $out/b_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$out/b_$name.json")); print("%-28s" % "$name", d["ms_per_step"], d.get("host_issue_ms_per_step"), d.get("host_phase_ms_per_step"))
except Exception as e: print("$name failed", e); print(open("$out/b_$name.err").read()[-1500:])
PY
}
run mt_inline MARIUS_MT_PREFETCH=0 MARIUS_SHARDED_FINE=1
run mt_inline_exact MARIUS_MT_PREFETCH=0 MARIUS_SHARDED_FINE=1 MARIUS_EXCHANGE=exact
run default MARIUS_SHARDED_FINE=1
