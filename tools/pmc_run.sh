#!/bin/bash
# usage: tools/pmc_run.sh <tag> <counters...>   (separate --pmc pass, kernel-trace only; see MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --pmc $@ -d gpurun_out/pmc_$tag -o pmc --output-format csv -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
ls gpurun_out/pmc_$tag | head -3
