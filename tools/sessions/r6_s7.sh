#!/bin/bash
# round 6 session 7: MT fill back to 256 threads; the driver's command with the map chain fused / unfused; one trajectory test
tag=${1:-r6s7}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout -s KILL 300 python -m pytest tests/test_gpu_host.py -q -m gpu -p no:cacheprovider -k "thousandfold" > $out/pytest_traj.txt 2>&1; echo "traj rc=$?"; grep -E "worst error|^FAILED|passed|failed" $out/pytest_traj.txt | cut -c1-250 | tail -30
for m in unfused fused; do
  MARIUS_MAPS=$m timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_$m.json 2> $out/bench_driver_$m.err
  MARIUS_MAPS=$m timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-arith-check --no-cpu-baseline --no-fp32-pass > $out/bench_100_$m.json 2> $out/bench_100_$m.err
  MARIUS_MAPS=$m MARIUS_FORCE_SHARDED=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-arith-check > $out/bench_sharded_w1_$m.json 2> $out/bench_sharded_w1_$m.err
done
python - <<PY
import json
for f in ("bench_driver_unfused", "bench_driver_fused", "bench_100_unfused", "bench_100_fused", "bench_sharded_w1_unfused", "bench_sharded_w1_fused"):
    try:
        d=json.load(open("$out/%s.json" % f)); print(f, d["ms_per_step"], d.get("host_busy_ms_per_step"), {k: v["avg_ms"] for k, v in d.get("kernels", {}).items() if k in ("sort_unique", "mt19937_fill")})
    except Exception as e: print(f, "failed", e); print(open("$out/%s.err" % f).read()[-1500:])
PY
