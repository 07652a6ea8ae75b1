#!/bin/bash
# tests/test_gpu_host.py as a whole (the sharded-trainer tests 48 tests into the process), repeated; first failure's message kept
# usage (GPU box): bash tools/sessions/r5_host_file_loop.sh <tag> [iterations] [seconds per iteration]
tag=${1:-r5hostloop}; n=${2:-2}; lim=${3:-85}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
export TORCH_NCCL_HEARTBEAT_TIMEOUT_SEC=60
for i in $(seq 1 $n); do
  s=$(date +%s)
  timeout -s KILL $lim python -u -m pytest tests/test_gpu_host.py -q -m gpu -x --tb=short -p no:cacheprovider -o faulthandler_timeout=50 > $out/it_$i.log 2>&1
  rc=$?
  echo "iteration $i rc=$rc $(( $(date +%s) - s )) s: $(tail -1 $out/it_$i.log | cut -c1-120)"
  if [ $rc -ne 0 ]; then tail -40 $out/it_$i.log | cut -c1-300; break; fi
done
