"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, separate runs) into profiles/<out>.json.
usage: python tools/pmc_traffic.py <fetch_counter_csv> <write_counter_csv> <out.json>
hbm_bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 correction)."""
import csv, json, re, sys, collections

def fold(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("marius::", "")
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items() if v[1]}

f = fold(sys.argv[1], "FETCH_SIZE")
w = fold(sys.argv[2], "WRITE_SIZE")
out = {"_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) of `python bench.py --steps 4 --warmup 2 --no-cpu-baseline` "
                   "on MI355X, Freebase86m workload; averages per launch. hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled per "
                   "MI355X_MICROARCH.md (gfx950 rocprofv3 reports 1/2 of a wide coalesced read); WRITE_SIZE uncalibrated.",
       "kernels": {}}
for k in sorted(set(f) | set(w)):
    if not (k.startswith("lp_") or k.startswith("gather") or k.startswith("seg_") or k.startswith("adagrad") or "kernel" in k and "at::" not in k and "rocprim" not in k):
        continue
    fs, ws = f.get(k, 0.0), w.get(k, 0.0)
    out["kernels"][k] = {"FETCH_SIZE_KB": round(fs, 1), "WRITE_SIZE_KB": round(ws, 1), "hbm_bytes": int((2 * fs + ws) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out["kernels"]), "kernels")
