#!/bin/bash
# round 5, session 7: forced-sharded world 1 — CUs left free by the persistent flash launches for the exchange / preparation streams
tag=${1:-s7}
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
    d=json.load(open("$out/b_$name.json")); print("%-28s" % "$name", d["ms_per_step"], d.get("host_issue_ms_per_step"), d["roofline"]["avg_ms"])
except Exception as e: print("$name failed", e); print(open("$out/b_$name.err").read()[-1500:])
PY
}
for r in 0 16 32 48 64 96; do run fixed_reserve$r MARIUS_FLASH_RESERVE=$r; done
for r in 0 32 64; do run exact_reserve$r MARIUS_FLASH_RESERVE=$r MARIUS_EXCHANGE=exact; done
run fixed_reserve32_stale2 MARIUS_FLASH_RESERVE=32 MARIUS_SHARDED_STALENESS=2
