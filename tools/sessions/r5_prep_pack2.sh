#!/bin/bash
# granularity of the fused prep + pack launch: MARIUS_PREP_PACK_GROUP in {1, 32, 256, 2048, 1000000 (= prep first)} against the two launches,
# 20-step driver command each, two rounds; one-step timeline of the prep-first form.   usage (GPU box): bash tools/sessions/r5_prep_pack2.sh <tag>
tag=${1:-r5pp2}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_gpu_flash.py -x -q -m gpu -k "one_launch" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
MARIUS_PREP_PACK_GROUP=100 timeout 600 python -m pytest tests/test_gpu_flash.py -x -q -m gpu -k "one_launch" > $out/pytest_g100.log 2>&1; tail -2 $out/pytest_g100.log
B="--no-cpu-baseline --no-fp32-pass --no-arith-check --no-profile"
for i in 1 2; do
  MARIUS_PREP_PACK_FUSED=0 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $out/two_$i.json 2> $out/two_$i.err
  for g in 1 32 256 2048 1000000; do
    MARIUS_PREP_PACK_GROUP=$g timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $out/g${g}_$i.json 2> $out/g${g}_$i.err
  done
done
MARIUS_PREP_PACK_GROUP=1000000 bash tools/sessions/gpu_session_timeline.sh ${tag}_tl > /dev/null 2>&1; cp gpurun_out/${tag}_tl/timeline.txt $out/timeline_prep_first.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.load(open(f)); print(f.split('/')[-1], d["ms_per_step"])
    except Exception as e: print(f, "FAILED", e)
PY
head -6 $out/timeline_prep_first.txt | cut -c1-140
