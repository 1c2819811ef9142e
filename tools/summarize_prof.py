#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/prof_<tag>{,_fetch,_write}) into the
small files committed under profiles/: kernel stats CSV, per-kernel PMC averages (JSON)."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]                      # e.g. r01b
src = os.path.join("gpurun_out", f"prof_{tag}")
out = "profiles"
os.makedirs(out, exist_ok=True)
ks = glob.glob(os.path.join(src, "*", "*_kernel_stats.csv"))
if ks:
    rows = list(csv.DictReader(open(ks[0])))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
# per (kernel, grid) durations from the trace: needed because one kernel runs at several sizes
tr = glob.glob(os.path.join(src, "*", "*_kernel_trace.csv"))
per = collections.defaultdict(list)
if tr:
    for r in csv.DictReader(open(tr[0])):
        per[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]), r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pmc = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    cc = glob.glob(os.path.join(f"{src}_{kind}", "*", "*_counter_collection.csv"))
    if not cc:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] == ctr:
            agg[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for (k, g), v in agg.items():
        pmc.setdefault(f"{k}@grid{g}", {})[ctr + "_KB_avg"] = sum(v) / len(v)
        pmc[f"{k}@grid{g}"]["launches_" + kind] = len(v)
summary = {"tag": tag, "kernels": [
    {"kernel": k, "grid_threads": g, "vgpr": vg, "lds": lds, "scratch": sc, "launches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}
    for (k, g, vg, lds, sc), v in sorted(per.items(), key=lambda kv: -sum(kv[1]))], "pmc": pmc,
    "note": "PMC units as reported by rocprofv3 (KB). gfx950: FETCH_SIZE under-reports wide coalesced streaming reads by 2x (MI355X_MICROARCH.md); 64-B gathers are uncalibrated. Collected in separate --pmc passes of the same bench command."}
json.dump(summary, open(os.path.join(out, f"{tag}_summary.json"), "w"), indent=1)
print("wrote", out, tag, len(per), "kernel configs")
