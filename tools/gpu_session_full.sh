#!/bin/bash
# whole GPU suite + smoke + bench lines (default, degree_fraction 0.5)
tag=${1:-full}
ulimit -c 0
mkdir -p gpurun_out/$tag
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/$tag/pytest_gpu.txt 2>&1; tail -6 gpurun_out/$tag/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --steps 100 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
timeout 300 python bench.py --steps 100 --degree-fraction 0.5 --no-cpu-baseline --no-fp32-pass > gpurun_out/$tag/bench_f05.json 2> gpurun_out/$tag/bench_f05.err
python - <<PY
import json
for f in ("bench","bench_f05"):
    try:
        d=json.load(open("gpurun_out/$tag/%s.json"%f))
        print(f, d["ms_per_step"], d["value"], d["loss_last_batch"], {k:(v["avg_ms"]) for k,v in d["kernels"].items()})
    except Exception as e:
        print(f, "FAILED", e); print(open("gpurun_out/$tag/%s.err"%f).read()[-1500:])
PY
echo "== forced-sharded world 1"
MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline > gpurun_out/$tag/bench_sharded_w1.json 2> gpurun_out/$tag/bench_sharded_w1.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$tag/bench_sharded_w1.json"))
    print("sharded w1", d["ms_per_step"], d.get("device_span_ms"), d.get("host_issue_ms_per_step"))
except Exception as e:
    print("sharded FAILED", e); print(open("gpurun_out/$tag/bench_sharded_w1.err").read()[-1500:])
PY
