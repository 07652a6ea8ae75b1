#!/bin/bash
# same-box A/B of the fused prep + pack launch (default) against the two launches (MARIUS_PREP_PACK_FUSED=0): the new bit-equality test, the driver's
# bench command alternated a/b/a/b, 100-step runs, one-step timeline of the default.   usage (GPU box): bash tools/sessions/r5_prep_pack.sh <tag>
tag=${1:-r5pp}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_gpu_flash.py -x -q -m gpu -k "one_launch or train_step or deterministic" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
B="--no-cpu-baseline --no-fp32-pass --no-arith-check --no-profile"
for i in 1 2; do
  MARIUS_PREP_PACK_FUSED=0 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $out/two_$i.json 2> $out/two_$i.err
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $out/one_$i.json 2> $out/one_$i.err
done
MARIUS_PREP_PACK_FUSED=0 timeout 200 python bench.py --steps 100 $B > $out/two_100.json 2> $out/two_100.err
timeout 200 python bench.py --steps 100 $B > $out/one_100.json 2> $out/one_100.err
bash tools/sessions/gpu_session_timeline.sh ${tag}_tl > /dev/null 2>&1; cp gpurun_out/${tag}_tl/timeline.txt $out/timeline_one_step.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.load(open(f)); print(f.split('/')[-1], d["ms_per_step"])
    except Exception as e: print(f, "FAILED", e)
PY
head -12 $out/timeline_one_step.txt | cut -c1-140
