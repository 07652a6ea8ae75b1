"""Per-step kernel table (per queue) from a rocprofv3 kernel_trace.csv: python tools/trace_kernel_table.py <kernel_trace.csv> [anchor-kernel-substring]
steps = occurrences of the anchor kernel; per kernel and queue: calls per step, busy us per step, average us."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "lp_prep2"
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
anch = sorted(r["s"] for r in rows if anchor in r["Kernel_Name"])
# steady-state window: skip the first and last quarter of the anchor occurrences
lo, hi = anch[len(anch) // 4], anch[3 * len(anch) // 4]
nsteps = 3 * len(anch) // 4 - len(anch) // 4
agg = collections.defaultdict(lambda: [0, 0])
busy = collections.defaultdict(int)
for r in rows:
    if lo <= r["s"] < hi:
        k = (r[qkey], r["Kernel_Name"][:110])
        agg[k][0] += 1
        agg[k][1] += r["e"] - r["s"]
        busy[r[qkey]] += r["e"] - r["s"]
print("window: %d steps, %.1f us per step" % (nsteps, (hi - lo) / 1e3 / nsteps))
for q in sorted(busy, key=lambda q: -busy[q]):
    print("queue %s busy %.1f us per step" % (q, busy[q] / 1e3 / nsteps))
print("%-4s %-110s %8s %10s %9s" % ("q", "kernel", "calls/st", "us/step", "avg us"))
for (q, name), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-4s %-110s %8.2f %10.1f %9.2f" % (q, name, n / nsteps, t / 1e3 / nsteps, t / 1e3 / n))
