#!/bin/bash
# round 6 session 10: per-kernel time of early vs late training steps (rocprofv3 totals at 60 and 260 steps)
tag=${1:-r6s10}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $R
for n in 60 260; do
  TRAIN_STEPS=$n timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof$n -o t --output-format csv -- python tools/train_n.py 2>&1 | grep "^steps" 
  f=$(find $out/prof$n -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_$n.csv; rm -rf $out/prof$n
done
python - <<PY
import csv
def load(n):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open("$out/kernel_stats_%d.csv" % n))}
a, b = load(60), load(260)
rows = []
for k, (cb, tb) in b.items():
    ca, ta = a.get(k, (0, 0.0))
    if cb - ca <= 0 or ca == 0: continue
    rows.append((k, ta / ca / 1e3, (tb - ta) / (cb - ca) / 1e3, ca / 60.0, ta / 60 / 1e3, (tb - ta) / 200 / 1e3))
rows.sort(key=lambda r: -r[5])
print("%-70s %9s %9s %6s %10s %10s" % ("kernel", "avg early", "avg late", "calls", "us/step e", "us/step l"))
for r in rows[:32]: print("%-70s %9.2f %9.2f %6.2f %10.1f %10.1f" % (r[0][:70], r[1], r[2], r[3], r[4], r[5]))
print("sum us/step early %.1f late %.1f" % (sum(r[4] for r in rows), sum(r[5] for r in rows)))
PY
