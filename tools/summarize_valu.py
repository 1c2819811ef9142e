#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_valu (rocprofv3 --pmc VALUBusy VALUUtilization) -> profiles/<tag>_valu.json"""
import collections, csv, glob, json, sqlite3, sys
tag = sys.argv[1]
found = glob.glob(f"gpurun_out/prof_{tag}_valu/*/*_counter_collection.csv")
if found:
    rows = list(csv.DictReader(open(found[0])))
else:   # ROCm 7: rocprofv3 writes a SQLite database; same fields through its counters_collection view
    rows = []
    for path in glob.glob(f"gpurun_out/prof_{tag}_valu/*/*_results.db"):
        cur = sqlite3.connect(path).cursor()
        cur.execute("select kernel_name, grid_size, queue_id, counter_name, value, start from counters_collection")
        rows += [{"Kernel_Name": a, "Grid_Size": str(b), "Queue_Id": c, "Counter_Name": d, "Counter_Value": e, "Start_Timestamp": f} for a, b, c, d, e, f in cur.fetchall()]
per_k, per_g = collections.defaultdict(lambda: collections.defaultdict(list)), collections.defaultdict(lambda: collections.defaultdict(list))
last_digits = {}
# round 5: the library's own launch log of the profiled run (bench.py --acc-log): record i = the i-th msm_accumulate launch
import os
log_path = f"gpurun_out/prof_{tag}_valu_acclog.json"
log = json.load(open(log_path))["launches"] if os.path.exists(log_path) else None
rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
first_counter = rows[0]["Counter_Name"] if rows else None
n_acc = sum(1 for r in rows if r["Counter_Name"] == first_counter and r["Kernel_Name"].split("(")[0].replace("void ", "") == "sg::msm_accumulate")
exact = log is not None and 0 < len(log) <= n_acc
seen = -1
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if name == "sg::msm_digits":
        last_digits[r["Queue_Id"]] = r["Grid_Size"]
    per_k[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    per_g[f"{name}@grid{r['Grid_Size']}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if name == "sg::msm_accumulate":   # launches of different jobs can share a grid: also keyed by the job
        if r["Counter_Name"] == first_counter:
            seen += 1
        if exact and seen < len(log):
            rec = log[seen]
            job = rec["n"] if rec["M"] == 1 else f"{rec['M']}x{rec['n']}"
        else:
            job = last_digits.get(r["Queue_Id"], 0)
        per_g[f"{name}@grid{r['Grid_Size']}@job{job}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = lambda d: {f"{c}_avg": round(sum(v) / len(v), 2) for c, v in d.items()}
main = ("msm_accumulate", "ntt_pass", "gates_kernel", "quot_", "mst_", "msm_reduce")
out = {"tag": tag, "msm_accumulate_attribution": "library launch log (exact)" if exact else "preceding msm_digits on the queue (heuristic)", "note": "rocprofv3 --pmc VALUBusy VALUUtilization over the default bench.py run (separate pass): VALUBusy = % of cycles "
                           "the vector ALUs are busy, VALUUtilization = % of active lanes",
       "kernels": {k: dict(avg(v), launches=len(next(iter(v.values())))) for k, v in sorted(per_k.items())},
       "per_grid": {k: avg(v) for k, v in sorted(per_g.items()) if any(m in k for m in main)}}
json.dump(out, open(f"profiles/{tag}_valu.json", "w"), indent=1)
print("wrote", f"profiles/{tag}_valu.json", len(out["kernels"]), "kernels")
