#!/bin/bash
# whole GPU suite (all failures listed, not -x) + the printed arithmetic pairs + smoke + bench lines (default, forced-sharded world 1)
# usage (GPU box): bash tools/sessions/gpu_session_full.sh <tag>   -> gpurun_out/<tag>/
tag=${1:-full}
ulimit -c 0
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/$tag/pytest_gpu.txt 2>&1; tail -15 gpurun_out/$tag/pytest_gpu.txt | cut -c1-300
grep -E "^(FAILED|ERROR)" gpurun_out/$tag/pytest_gpu.txt | cut -c1-250 | head -40
timeout 600 python -m pytest tests/test_gpu_flash.py -q -m gpu -s -k "arithmetic_against" -p no:cacheprovider > gpurun_out/$tag/arith.txt 2>&1; grep -E "device max|passed|failed|ratio" gpurun_out/$tag/arith.txt | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 100 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python - <<PY
import json
for f in ("bench",):
    try:
        d=json.load(open("gpurun_out/$tag/%s.json"%f))
        print(f, d["ms_per_step"], d["value"], d["dtype"], d["loss_last_batch"], {k:(v["avg_ms"]) for k,v in d["kernels"].items()})
        print("arith ok:", d["arith_check"] and d["arith_check"]["ok"], d["arith_check"] and d["arith_check"]["seconds"], "fp32_exact", d["fp32_exact"] and d["fp32_exact"]["ms_per_step"], "fast_path", d["fast_path"] and d["fast_path"]["ms_per_step"])
    except Exception as e:
        print(f, "FAILED", e); print(open("gpurun_out/$tag/%s.err"%f).read()[-2500:])
PY
echo "== forced-sharded world 1"
MARIUS_FORCE_SHARDED=1 timeout 400 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > gpurun_out/$tag/bench_sharded_w1.json 2> gpurun_out/$tag/bench_sharded_w1.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$tag/bench_sharded_w1.json"))
    print("sharded w1", d["ms_per_step"], d["dtype"], d.get("host_issue_ms_per_step"), d.get("host_phase_ms_per_step"))
except Exception as e:
    print("sharded FAILED", e); print(open("gpurun_out/$tag/bench_sharded_w1.err").read()[-2500:])
PY
