#!/bin/bash
# the whole -m gpu suite, unbuffered, stopping at the first failure with its traceback; stacks of every thread of a test that takes > 90 s; hard limits
# usage (GPU box): bash tools/sessions/r5_suite_diag.sh <tag>
tag=${1:-r5diag}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
export TORCH_NCCL_HEARTBEAT_TIMEOUT_SEC=120
s=$(date +%s)
timeout -s KILL 420 python -u -m pytest tests -q -m gpu -x --tb=short -p no:cacheprovider -o faulthandler_timeout=90 --timeout 150 --timeout-method=thread > $out/pytest_gpu.txt 2>&1
echo "rc=$? $(( $(date +%s) - s )) s"
tail -5 $out/pytest_gpu.txt | cut -c1-300
grep -n "Error\|error\|Exception\|FAILED\|Timeout" $out/pytest_gpu.txt | head -20 | cut -c1-300
