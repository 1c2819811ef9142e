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
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if name == "sg::msm_digits":
        last_digits[r["Queue_Id"]] = r["Grid_Size"]
    per_k[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    per_g[f"{name}@grid{r['Grid_Size']}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if name == "sg::msm_accumulate":   # launches of different jobs can share a grid: also keyed by the job shape
        per_g[f"{name}@grid{r['Grid_Size']}@job{last_digits.get(r['Queue_Id'], 0)}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = lambda d: {f"{c}_avg": round(sum(v) / len(v), 2) for c, v in d.items()}
main = ("msm_accumulate", "ntt_pass", "gates_kernel", "quot_", "mst_", "msm_reduce")
out = {"tag": tag, "note": "rocprofv3 --pmc VALUBusy VALUUtilization over the default bench.py run (separate pass): VALUBusy = % of cycles "
                           "the vector ALUs are busy, VALUUtilization = % of active lanes",
       "kernels": {k: dict(avg(v), launches=len(next(iter(v.values())))) for k, v in sorted(per_k.items())},
       "per_grid": {k: avg(v) for k, v in sorted(per_g.items()) if any(m in k for m in main)}}
json.dump(out, open(f"profiles/{tag}_valu.json", "w"), indent=1)
print("wrote", f"profiles/{tag}_valu.json", len(out["kernels"]), "kernels")
