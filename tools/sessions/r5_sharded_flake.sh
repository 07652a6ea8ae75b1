#!/bin/bash
# hunt for an intermittent failure of the sharded-trainer tests: the same tests in fresh processes until one fails (each under a hard timeout)
# usage (GPU box): bash tools/sessions/r5_sharded_flake.sh <tag> [iterations]
tag=${1:-r5flake}; n=${2:-10}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
export TORCH_NCCL_HEARTBEAT_TIMEOUT_SEC=60
for i in $(seq 1 $n); do
  s=$(date +%s)
  timeout -s KILL 150 python -X faulthandler -m pytest tests/test_gpu_host.py -q -m gpu -k "sharded_trainer" -x --timeout 100 -p no:cacheprovider > $out/it_$i.log 2>&1
  rc=$?
  echo "iteration $i rc=$rc $(( $(date +%s) - s )) s: $(tail -1 $out/it_$i.log | cut -c1-120)"
  if [ $rc -ne 0 ]; then tail -60 $out/it_$i.log | cut -c1-300; break; fi
  rm -f $out/it_$i.log
done
