#!/bin/bash
# what to run first if GPU use is reopened: the whole suite (all failures listed, not -x), smoke, the driver's command twice, then the round's evidence
tag=${1:-final}
ulimit -c 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt | cut -c1-300
grep -E "^(FAILED|ERROR)" $out/pytest_gpu.txt | cut -c1-300 | head -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt | cut -c1-200
for i in 1 2; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_$i.json 2> $out/bench_driver_$i.err
  python -c "
import json
d=json.load(open('$out/bench_driver_$i.json')); print('driver cmd', d['ms_per_step'], 'arith ok', d['arith_check']['ok'], 'frac', d['roofline']['frac'])" 2>&1 | tail -1
done
bash tools/sessions/evidence.sh ${tag}_ev 2>&1 | tail -20
