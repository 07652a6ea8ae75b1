#!/bin/bash
# usage: tools/pmc_bench.sh <tag> <counter>   — one rocprofv3 --pmc pass (kernel-trace only) over the default bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --pmc $@ -d gpurun_out/pmc_$tag -o pmc --output-format csv -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/pmc_$tag.log 2>&1
ls gpurun_out/pmc_$tag | head -3
