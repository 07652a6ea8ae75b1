"""Print a per-step kernel table from a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> <steps_total>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = 0
print("%-84s %7s %10s %9s" % ("kernel", "calls", "us/step", "avg us"))
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-84s %7s %10.1f %9.2f" % (r["Name"][:84], r["Calls"], float(r["TotalDurationNs"]) / 1e3 / n, float(r["AverageNs"]) / 1e3))
print("sum us/step", sum(float(r["TotalDurationNs"]) for r in rows) / 1e3 / n)
