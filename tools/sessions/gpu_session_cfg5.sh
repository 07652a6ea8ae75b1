#!/bin/bash
# cfg5 at Twitter-2010's real size: 41,652,230 nodes x d = 400 (66.6 GB + 66.6 GB Adagrad state), 1.47 B synthetic edges, 16 partitions of which 8 resident.
# usage (GPU box): bash tools/sessions/gpu_session_cfg5.sh <tag> [edges]    -> gpurun_out/<tag>/{partition_cfg5.json, bench_twitter.json}
tag=${1:-cfg5}; edges=${2:-1470000000}
ulimit -c 0
out=gpurun_out/$tag; mkdir -p $out
df -BG /dev/shm | tail -1; free -g | head -2
avail=$(df -BG --output=avail /dev/shm | tail -1 | tr -dc 0-9)
if [ "$avail" -lt 150 ]; then echo "/dev/shm too small for 133 GB of partition files: $avail GB"; exit 0; fi
timeout 120 python bench.py --workload twitter --no-arith-check --no-cpu-baseline > $out/bench_twitter.json 2> $out/bench_twitter.err
python -c "
import json; d=json.load(open('$out/bench_twitter.json')); print('twitter in-memory step', d['ms_per_step'], d['positive_edges_per_s'], d['dtype'][:70], d['roofline'])" || tail -3 $out/bench_twitter.err
PB_TRACE=1 PB_TRACE_DUMP_AFTER=600 timeout 800 python tools/bench_partition_train.py --nodes 41652230 --edges $edges --partitions 16 --capacity 8 --d 400 --relations 1 --dir /dev/shm --skip-device-memory > $out/partition_cfg5.json 2> $out/partition_cfg5.err
tail -c 1500 $out/partition_cfg5.json; echo; grep -E "^\[" $out/partition_cfg5.err | tail -8
rm -f /dev/shm/pb_bench_*.bin
