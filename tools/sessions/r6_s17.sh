#!/bin/bash
# round 6 session 17: negative pack inside the prep launch: A/B in one lease, flash / fullshape / host tests
tag=${1:-r6s17}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
run() { name=$1; shift; timeout 900 python bench.py --gpus 1 --no-cpu-baseline --no-fp32-pass --no-arith-check "$@" > $out/$name.json 2> $out/$name.err; python -c "
import json
d=json.load(open('$out/$name.json')); k=d['kernels']; print('$name', d['ms_per_step'], 'prep', k['lp_prep']['avg_ms'], k['lp_prep']['frac'], 'pack', k.get('lp_pack',{}).get('avg_ms'))" 2>&1 | tail -1; }
for i in 1 2; do
MARIUS_PREP_PACK=split run split100_$i --steps 100 --warmup 10
run merged100_$i --steps 100 --warmup 10
done
MARIUS_PREP_PACK=split run split20 --steps 20 --warmup 5
run merged20 --steps 20 --warmup 5
timeout 1500 python -m pytest tests/test_gpu_flash.py tests/test_gpu_fullshape.py tests/test_gpu_host.py -q -m gpu -p no:cacheprovider -x > $out/pytest.txt 2>&1; grep -E "passed|failed" $out/pytest.txt | tail -2 | cut -c1-200; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-300 | head
