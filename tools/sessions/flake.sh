#!/bin/bash
# the sharded-trainer tests in FRESH processes, one after the other, each under a hard timeout; stops at the first failure and prints its tail
# usage (GPU box): bash tools/sessions/flake.sh <tag> [iterations] [pytest -k expression]
tag=${1:-flake}; n=${2:-50}; k=${3:-"world1_equals_synchronous_trainer"}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
export TORCH_NCCL_HEARTBEAT_TIMEOUT_SEC=60
ok=0
for i in $(seq 1 $n); do
  s=$(date +%s)
  timeout -s KILL 200 python -X faulthandler -m pytest tests/test_gpu_host.py -q -m gpu -k "$k" -x --timeout 150 -p no:cacheprovider > $out/it_$i.log 2>&1
  rc=$?
  echo "iteration $i rc=$rc $(( $(date +%s) - s )) s: $(tail -1 $out/it_$i.log | cut -c1-120)" >> $out/summary.txt
  if [ $rc -ne 0 ]; then tail -80 $out/it_$i.log | cut -c1-400; break; fi
  ok=$((ok+1))
  rm -f $out/it_$i.log
done
echo "flake loop '$k': $ok of $n fresh processes passed" | tee -a $out/summary.txt
