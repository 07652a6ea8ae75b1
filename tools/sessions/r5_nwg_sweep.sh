#!/bin/bash
# workgroup count of the persistent flash launches re-checked after this round's kernel changes (the rule picks 480 of 512 at the bench shape)
# usage (GPU box): bash tools/sessions/r5_nwg_sweep.sh <tag>
tag=${1:-r5nwg}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
B="--no-cpu-baseline --no-fp32-pass --no-arith-check --no-profile"
for i in 1 2; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $out/rule_$i.json 2> $out/rule_$i.err
  for n in 400 416 448 464 480 496 512; do
    MARIUS_FLASH_NWG=$n timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $out/n${n}_$i.json 2> $out/n${n}_$i.err
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["ms_per_step"])
    except Exception as e: print(f, "FAILED", e)
PY
