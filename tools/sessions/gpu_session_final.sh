#!/bin/bash
# end-of-round evidence: whole GPU suite, smoke, default bench line, kernel stats + PMC traffic + one-step timeline of the default command,
# forced-sharded world-1 line, cfg5-shaped (twitter) line.   usage (GPU box): bash tools/sessions/gpu_session_final.sh <tag>
tag=${1:-final}
bash tools/sessions/gpu_session_full.sh $tag
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
