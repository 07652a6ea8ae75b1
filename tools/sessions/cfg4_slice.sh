#!/bin/bash
# round 6 session 18: cfg4 slice (neighbour sampling + GraphSage aggregation) parity tests; layout-config check
tag=${1:-r6s18}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1500 python -m pytest tests/test_gpu_zz_unverified_cfg4.py -q -m gpu -p no:cacheprovider -x > $out/pytest_nbr.txt 2>&1; grep -E "passed|failed|Error" $out/pytest_nbr.txt | tail -5 | cut -c1-300; grep -E "^(FAILED|ERROR)" $out/pytest_nbr.txt | cut -c1-300 | head
timeout 600 python -m pytest tests/test_gpu_zz_unverified_cfg4.py -q -m gpu -p no:cacheprovider -k "layout_refuses" > $out/pytest_cfg.txt 2>&1; tail -3 $out/pytest_cfg.txt | cut -c1-300
