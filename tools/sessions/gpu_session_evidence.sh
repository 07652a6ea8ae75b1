#!/bin/bash
# profile + timeline + the twitter / degree-fraction lines of the final code (the second half of tools/gpu_session_final.sh):  bash tools/sessions/gpu_session_evidence.sh
tag=r4fin2; mkdir -p gpurun_out/$tag
bash tools/sessions/gpu_session_profile.sh ${tag}_prof
bash tools/sessions/gpu_session_timeline.sh ${tag}_tl
timeout 200 python bench.py --workload twitter --no-arith-check --no-cpu-baseline > gpurun_out/$tag/bench_twitter.json 2> gpurun_out/$tag/bench_twitter.err
timeout 200 python bench.py --steps 100 --degree-fraction 0.5 --no-arith-check --no-cpu-baseline --no-fp32-pass > gpurun_out/$tag/bench_f05.json 2> gpurun_out/$tag/bench_f05.err
python - <<PY
import json
for f in ("bench_twitter","bench_f05"):
    try:
        d=json.load(open("gpurun_out/$tag/%s.json"%f)); print(f, d["ms_per_step"], d["positive_edges_per_s"])
    except Exception as e: print(f, "FAILED", e)
PY
