#!/bin/bash
# round 6 session 6: fused map launch with agent-scope accesses instead of fences; the three trajectory tests; gate rule; bench A/B
tag=${1:-r6s6}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
PM_TRIALS=3 timeout -s KILL 120 python -u tools/pm_debug.py 2>&1 | grep -v amdgpu.ids | tee $out/pm.txt | cut -c1-400 | tail -12
if ! grep -q "fused:" $out/pm.txt; then echo "fused launch not healthy: stopping"; exit 0; fi
timeout -s KILL 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "prepare_maps or sort_unique or segment_plan" > $out/pytest_maps.txt 2>&1; echo "maps rc=$?"; tail -3 $out/pytest_maps.txt | cut -c1-300
timeout -s KILL 600 python -m pytest tests/test_gpu_host.py -q -m gpu -p no:cacheprovider -k "trainer_epoch_matches or thousandfold or tracked_bound" > $out/pytest_traj.txt 2>&1; echo "traj rc=$?"; grep -E "worst error|^FAILED|passed|failed" $out/pytest_traj.txt | cut -c1-250 | tail -30
MARIUS_MAPS=unfused timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-arith-check --no-cpu-baseline --no-fp32-pass > $out/bench_unfused.json 2> $out/bench_unfused.err
timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-arith-check --no-cpu-baseline --no-fp32-pass > $out/bench_fused.json 2> $out/bench_fused.err
MARIUS_MAPS=unfused MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/bench_sharded_w1_unfused.json 2> $out/bench_sharded_w1_unfused.err
MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/bench_sharded_w1.json 2> $out/bench_sharded_w1.err
python - <<PY
import json
for f in ("bench_unfused", "bench_fused", "bench_sharded_w1_unfused", "bench_sharded_w1"):
    try:
        d=json.load(open("$out/%s.json" % f)); print(f, d["ms_per_step"], d.get("host_busy_ms_per_step"))
    except Exception as e: print(f, "failed", e); print(open("$out/%s.err" % f).read()[-1500:])
PY
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
python - <<PY
import json
try:
    d=json.load(open("$out/bench_driver.json")); a=d["arith_check"]; print("driver cmd", d["ms_per_step"], "arith ok", a["ok"], "seconds", a["seconds"]); print(json.dumps(a["verdict"])[:900])
    for p in a["per_input"]: print(p["input"][:40], p["ok"], p["equal_to_fp32_within_10pct"], p["worst_rms_vs_reference"], p["worst_max_vs_reference"])
    print({k: (v["avg_ms"]) for k, v in d["kernels"].items()})
except Exception as e: print("bench failed", e); print(open("$out/bench_driver.err").read()[-3000:])
PY
