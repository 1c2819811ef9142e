#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/prof_<tag>{,_fetch,_write}) into the
small files committed under profiles/: kernel stats CSV, per-kernel PMC averages (JSON)."""
import collections, csv, glob, json, os, sys
import sqlite3


def rocpd_rows(directory, view):
    """rows of a view of rocprofv3's SQLite output (ROCm 7: `<dir>/<host>/<pid>_results.db`), as dicts"""
    out = []
    # gpurun merges every run's database into the same directory: only the newest one is the run being summarised
    for path in sorted(glob.glob(os.path.join(directory, "*", "*_results.db")), key=os.path.getmtime)[-1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        cur.execute(f"select * from {view}")
        names = [d[0] for d in cur.description]
        out += [dict(zip(names, r)) for r in cur.fetchall()]
    return out


def acc_log_for(directory):
    """the library's own launch log of the profiled run (bench.py --acc-log <directory>_acclog.json): record i = the i-th
    msm_accumulate launch of the process in device order (the launches are chained), or None"""
    path = directory.rstrip("/") + "_acclog.json"
    if not os.path.exists(path):
        return None
    return json.load(open(path))["launches"]


def job_key(rec):
    """the job a launch belongs to, as the old keys named it (thread count of its msm_digits launch = n for one MSM)"""
    return rec["n"] if rec["M"] == 1 else f"{rec['M']}x{rec['n']}"


attribution = {}

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
elif glob.glob(os.path.join(src, "*", "*_results.db")):
    agg = collections.defaultdict(list)
    for r in rocpd_rows(src, "kernels"):
        agg[r["name"].split("(")[0]].append(r["duration"])
    total = sum(sum(v) for v in agg.values()) or 1
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 3), min(v), max(v)])
# per (kernel, grid) durations from the trace: needed because one kernel runs at several sizes
tr = glob.glob(os.path.join(src, "*", "*_kernel_trace.csv"))
per = collections.defaultdict(list)
# msm_accumulate launches of different jobs can share a grid (the launch is sized by an upper bound of the task count):
# they are told apart by the job shape, i.e. the thread count of the msm_digits launch that precedes them on the queue
acc_jobs = collections.defaultdict(list)
if tr:
    last_digits = {}
    for r in sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"])):
        name, grid = r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        per[(name, grid, r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])].append(dur)
        if name == "sg::msm_digits":
            last_digits[r["Queue_Id"]] = grid
        elif name == "sg::msm_accumulate":
            acc_jobs[(grid, last_digits.get(r["Queue_Id"], 0), None)].append(dur)
elif glob.glob(os.path.join(src, "*", "*_results.db")):
    last_digits = {}
    rows = sorted(rocpd_rows(src, "kernels"), key=lambda r: r["start"])
    log = acc_log_for(src)
    n_acc = sum(1 for r in rows if r["name"].split("(")[0] == "sg::msm_accumulate")
    # the log covers the launches up to the moment it was written: record i = i-th launch for as many records as it holds (the
    # grids must agree, which is checked launch by launch below); launches beyond it fall back to the heuristic and say so
    exact = log is not None and 0 < len(log) <= n_acc
    attribution["kernel_trace"] = {"method": "library launch log (exact: record i = i-th chained launch)" if exact else "preceding msm_digits on the queue (heuristic)",
                                   "launches_in_trace": n_acc, "records_in_log": None if log is None else len(log),
                                   "launches_beyond_the_log": (n_acc - len(log)) if exact else None}
    seen = 0
    for r in rows:
        name, grid = r["name"].split("(")[0], int(r["grid_x"]) * int(r["grid_y"]) * int(r["grid_z"])
        dur = r["duration"] / 1e3
        per[(name, grid, str(r["vgpr_count"]), str(r["lds_size"]), str(r["scratch_size"]))].append(dur)
        if name == "sg::msm_digits":
            last_digits[r["queue_id"]] = grid
        elif name == "sg::msm_accumulate":
            if exact and seen < len(log):
                rec = log[seen]
                assert rec["threads"] == grid, (seen, rec, grid)     # the log and the trace describe the same launch
                acc_jobs[(grid, job_key(rec), rec["jobs_in_flight"])].append(dur)
            else:
                acc_jobs[(grid, last_digits.get(r["queue_id"], 0), None)].append(dur)
            seen += 1
pmc = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    if not glob.glob(os.path.join(f"{src}_{kind}", "*", "*_counter_collection.csv")) and glob.glob(os.path.join(f"{src}_{kind}", "*", "*_results.db")):
        agg = collections.defaultdict(list)
        last_digits = {}
        crow = sorted(rocpd_rows(f"{src}_{kind}", "counters_collection"), key=lambda r: r["start"])
        log = acc_log_for(f"{src}_{kind}")
        n_acc = sum(1 for r in crow if r["counter_name"] == ctr and r["kernel_name"].split("(")[0] == "sg::msm_accumulate")
        exact = log is not None and 0 < len(log) <= n_acc
        attribution[kind] = {"method": "library launch log (exact)" if exact else "preceding msm_digits on the queue (heuristic)",
                             "launches_in_pass": n_acc, "records_in_log": None if log is None else len(log)}
        seen = 0
        for r in crow:
            name, grid = r["kernel_name"].split("(")[0], int(r["grid_size"])
            if name == "sg::msm_digits":
                last_digits[r["queue_id"]] = grid
            if r["counter_name"] == ctr:
                agg[(name, grid)].append(float(r["value"]))
                if name == "sg::msm_accumulate":
                    job = job_key(log[seen]) if (exact and seen < len(log)) else last_digits.get(r['queue_id'], 0)
                    agg[(name, f"{grid}@job{job}")].append(float(r["value"]))
                    seen += 1
        for (k, g), v in agg.items():
            pmc.setdefault(f"{k}@grid{g}", {})[ctr + "_KB_avg"] = sum(v) / len(v)
            pmc[f"{k}@grid{g}"]["launches_" + kind] = len(v)
        continue
    cc = glob.glob(os.path.join(f"{src}_{kind}", "*", "*_counter_collection.csv"))
    if not cc:
        continue
    agg = collections.defaultdict(list)
    last_digits = {}
    for r in sorted(csv.DictReader(open(cc[0])), key=lambda r: int(r["Start_Timestamp"])):
        name, grid = r["Kernel_Name"].split("(")[0], int(r["Grid_Size"])
        if name == "sg::msm_digits":
            last_digits[r["Queue_Id"]] = grid
        if r["Counter_Name"] == ctr:
            agg[(name, grid)].append(float(r["Counter_Value"]))
            if name == "sg::msm_accumulate":
                agg[(name, f"{grid}@job{last_digits.get(r['Queue_Id'], 0)}")].append(float(r["Counter_Value"]))
    for (k, g), v in agg.items():
        pmc.setdefault(f"{k}@grid{g}", {})[ctr + "_KB_avg"] = sum(v) / len(v)
        pmc[f"{k}@grid{g}"]["launches_" + kind] = len(v)
summary = {"tag": tag, "kernels": [
    {"kernel": k, "grid_threads": g, "vgpr": vg, "lds": lds, "scratch": sc, "launches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}
    for (k, g, vg, lds, sc), v in sorted(per.items(), key=lambda kv: -sum(kv[1]))],
    "msm_accumulate_by_job": [{"grid_threads": g, "job_threads": j, "jobs_in_flight_at_issue": f, "launches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}
                              for (g, j, f), v in sorted(acc_jobs.items(), key=lambda kv: -sum(kv[1]))],
    "msm_accumulate_attribution": attribution,
    "pmc": pmc,
    "note": "PMC units as reported by rocprofv3 (KB). gfx950: FETCH_SIZE under-reports wide coalesced streaming reads by 2x (MI355X_MICROARCH.md); 64-B gathers are uncalibrated. Collected in separate --pmc passes of the same bench command."}
json.dump(summary, open(os.path.join(out, f"{tag}_summary.json"), "w"), indent=1)
print("wrote", out, tag, len(per), "kernel configs")
