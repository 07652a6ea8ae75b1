#!/bin/bash
# round 5, session 10: folded column tail (d = 36 / 68 / 100) — flash + full-shape parity, then A/B against the seven-k-step layout on one box
tag=${1:-s10}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1200 python -m pytest tests/test_gpu_flash.py tests/test_gpu_fullshape.py -q -m gpu -x -p no:cacheprovider > $out/pytest.txt 2>&1; tail -6 $out/pytest.txt | cut -c1-300
grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-250 | head
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 100 --no-arith-check --no-cpu-baseline --no-fp32-pass > $out/b_$name.json 2> $out/b_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$out/b_$name.json")); print("%-12s" % "$name", d["ms_per_step"], d["loss_last_batch"], {k:v["avg_ms"] for k,v in d["kernels"].items() if k in ("lp_grad_adj","lp_grad_neg","lp_prep","lp_pack","lp_edge_bwd","segment_adagrad_scatter")})
except Exception as e: print("$name failed", e); print(open("$out/b_$name.err").read()[-1500:])
PY
}
for rep in 1 2; do
  run tail4_$rep A=1
  run k7_$rep MARIUS_FLASH_TAIL4=0
done
