#!/bin/bash
# round 6 session 5: fused map launch after the loop-shape fix: watchdog tool, map tests, suite, bench A/B
tag=${1:-r6s5}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
for n in 1 0; do
  echo "== MARIUS_PM_NWG=$n"
  MARIUS_PM_NWG=$n PM_TRIALS=2 timeout -s KILL 120 python -u tools/pm_debug.py 2>&1 | grep -v amdgpu.ids | tee $out/pm_nwg$n.txt | cut -c1-400 | tail -12
done
if ! grep -q "fused:" $out/pm_nwg0.txt; then echo "fused launch still not healthy: stopping"; exit 0; fi
timeout -s KILL 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "prepare_maps or sort_unique" > $out/pytest_maps.txt 2>&1; echo "maps rc=$?"; tail -5 $out/pytest_maps.txt | cut -c1-300
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt | cut -c1-300
grep -E "^(FAILED|ERROR)" $out/pytest_gpu.txt | cut -c1-300 | head -60
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
python - <<PY
import json
try:
    d=json.load(open("$out/bench_driver.json")); a=d["arith_check"]; print("driver cmd", d["ms_per_step"], "arith ok", a["ok"], "seconds", a["seconds"]); print(json.dumps(a["verdict"])[:600])
    for p in a["per_input"]: print(p["input"][:40], p["ok"], p["worst_rms_vs_reference"], p["worst_max_vs_reference"])
    print({k: (v["avg_ms"]) for k, v in d["kernels"].items()})
except Exception as e: print("bench failed", e); print(open("$out/bench_driver.err").read()[-3000:])
PY
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
