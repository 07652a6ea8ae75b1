#!/bin/bash
# the loader's gate event bound to the edge backward's completion (marius_lp_desc.bwd_done_event, default) against a hipEventRecord behind it
# (MARIUS_GATE_BOUND=0): tests, the driver's command alternated, one-step timeline.   usage (GPU box): bash tools/sessions/r5_gate_bound.sh <tag>
tag=${1:-r5gb}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_gpu_flash.py tests/test_gpu_host.py -x -q -m gpu -k "done_event or trainer or epoch" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
B="--no-cpu-baseline --no-fp32-pass --no-arith-check --no-profile"
for i in 1 2 3; do
  MARIUS_GATE_BOUND=0 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $out/record_$i.json 2> $out/record_$i.err
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $out/bound_$i.json 2> $out/bound_$i.err
done
MARIUS_GATE_BOUND=0 timeout 200 python bench.py --steps 100 $B > $out/record_100.json 2> $out/record_100.err
timeout 200 python bench.py --steps 100 $B > $out/bound_100.json 2> $out/bound_100.err
bash tools/sessions/gpu_session_timeline.sh ${tag}_tl > /dev/null 2>&1; cp gpurun_out/${tag}_tl/timeline.txt $out/timeline_one_step.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.load(open(f)); print(f.split('/')[-1], d["ms_per_step"], d.get("loss_last_batch"))
    except Exception as e: print(f, "FAILED", e)
PY
grep -n "gap" $out/timeline_one_step.txt | awk '$8+0 > 1.0' | cut -c1-140 | head
grep "q1" $out/timeline_one_step.txt | cut -c1-130
