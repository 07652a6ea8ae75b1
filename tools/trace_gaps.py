"""Per-stream timeline of one steady-state step from a rocprofv3 kernel_trace.csv:
python tools/trace_gaps.py <kernel_trace.csv> [anchor-kernel-substring] [which-occurrence]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "lp_prep_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
anchors = [r for r in rows if anchor in r["Kernel_Name"]]
if which < 0:
    which += len(anchors)
t0, t1 = anchors[which]["s"], anchors[which + 1]["s"]
print("step length us", (t1 - t0) / 1e3)
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
last = {}
for r in rows:
    if r["s"] < t0 or r["s"] >= t1:
        continue
    q = r[qkey]
    gap = (r["s"] - last[q]) / 1e3 if q in last else 0
    last[q] = r["e"]
    print("q%-3s +%8.1f us  dur %7.1f  gap %6.1f  %s" % (q, (r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, gap, r["Kernel_Name"][:70]))
