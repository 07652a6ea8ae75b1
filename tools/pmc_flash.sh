#!/bin/bash
# usage: tools/pmc_flash.sh <tag> <counters...>  — one --pmc pass (kernel-trace only) over tools/bench_flash.py, folded per kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --pmc $@ -d gpurun_out/pmcf_$tag -o pmc --output-format csv -- python tools/bench_flash.py > gpurun_out/pmcf_$tag.log 2>&1
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for p in glob.glob("gpurun_out/pmcf_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("marius::", "")
        if "flash" not in n and "lp_" not in n: continue
        a = acc[n][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for n in sorted(acc):
    print(n, {k: round(v[0] / v[1], 1) for k, v in acc[n].items()})
PY
