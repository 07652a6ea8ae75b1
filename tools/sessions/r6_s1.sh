#!/bin/bash
# round 6 session 1: new hand-off tests, the whole suite, driver bench, forced-sharded world 1, flake loop
tag=${1:-r6s1}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded2.py tests/test_gpu_host.py -q -m gpu -p no:cacheprovider -k "exchange_halves or overflow or sharded" > $out/pytest_new.txt 2>&1; tail -5 $out/pytest_new.txt | cut -c1-300
grep -E "^(FAILED|ERROR)" $out/pytest_new.txt | cut -c1-250 | head
bash tools/sessions/full.sh $tag
bash tools/sessions/flake.sh $tag/flake 50
