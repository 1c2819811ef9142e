#!/usr/bin/env python3
"""Joins the rocprofv3 passes of tools/prof_proof_r05.sh (one k = 17 proof of tools/create_proof_cpp, SG_PROVER_SERIAL=1) launch by
launch into one budget: per kernel of the proof -- launches, microseconds (kernel trace), VALU / SALU / LDS / VMEM wave-instructions,
VALU busy and lane utilisation, FETCH_SIZE, WRITE_SIZE, the wave-cycle split (parked / issue-stalled / issuing), the bytes the
launch must move as it is written (`min_bytes`: inputs once + outputs once, formula in `min_bytes_model`), and from those the
fraction of the HBM roofline and of the kernel's own issue floor.

    python tools/proof_budget.py <work dir with serial_trace, serial_insts, ...> <tag> <out.json>

A proof = the launches between two `count_noncanonical_kernel` launches (the first kernel of every proof); the LAST COMPLETE
untimed proof of each pass is taken.  The program is deterministic, so launch i of the proof is the same launch in every pass
(checked: same kernel name and grid; a pass whose sequence differs is joined by kernel name only and says so).
msm_accumulate launches are attributed to their jobs by the library's own launch log (SG_ACC_LOG, exact)."""
import collections
import glob
import json
import os
import sqlite3
import sys

HBM_PEAK = 8.0e12      # B/s, /opt/skills/guides/MI355X_MICROARCH.md
SIMDS = 1024           # 256 CUs x 4
K = 17
N = 1 << K
NE = 5 * N             # the quotient lives on five cosets
R = 32                 # bytes per field element


def rows_of(directory, view):
    paths = sorted(glob.glob(os.path.join(directory, "*", "*_results.db")), key=os.path.getmtime)
    if not paths:
        return None
    cur = sqlite3.connect(paths[-1]).cursor()
    cur.execute(f"select * from {view}")
    names = [d[0] for d in cur.description]
    return [dict(zip(names, r)) for r in cur.fetchall()]


def short(name):
    return name.split("(")[0].replace("void ", "")


MARKER = "sg::count_noncanonical_kernel"


def last_proof(seq):
    """seq: launches in device order, each a dict with `name`; returns the launches of the last complete untimed proof.
    create_proof_cpp runs reps + 1 untimed proofs and then one with per-phase synchronisation: the proof between the last two
    markers is the last untimed one."""
    marks = [i for i, d in enumerate(seq) if d["name"] == MARKER]
    if len(marks) < 3:
        raise SystemExit(f"only {len(marks)} proofs found")
    return seq[marks[-2]:marks[-1]], len(marks)


def trace_launches(directory):
    rows = rows_of(directory, "kernels")
    seq = [{"name": short(r["name"]), "grid": int(r["grid_x"]) * int(r["grid_y"]) * int(r["grid_z"]), "wg": int(r["workgroup_x"]) if "workgroup_x" in r else None,
            "us": r["duration"] / 1e3, "vgpr": r.get("vgpr_count"), "lds": r.get("lds_size"), "scratch": r.get("scratch_size")}
           for r in sorted(rows, key=lambda r: r["start"])]
    return seq


def pmc_launches(directory):
    """dispatches of a --pmc pass in device order: name, grid, {counter: value}.  A counter that comes back as several rows of one
    dispatch (one per dimension instance) is summed when it is a count and averaged when it is a derived percentage."""
    rows = rows_of(directory, "counters_collection")
    if rows is None:
        return None
    key = "dispatch_id" if rows and "dispatch_id" in rows[0] else None
    groups = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: (r["start"], r.get("dispatch_id", 0))):
        gid = (r[key] if key else (r["start"], r["kernel_name"]))
        g = groups.setdefault(gid, {"name": short(r["kernel_name"]), "grid": int(r["grid_size"]), "c": collections.defaultdict(list)})
        g["c"][r["counter_name"]].append(float(r["value"]))
    out = []
    for g in groups.values():
        vals = {}
        for cname, v in g["c"].items():
            derived = cname in ("VALUBusy", "VALUUtilization")
            vals[cname] = sum(v) / len(v) if derived else sum(v)
        out.append({"name": g["name"], "grid": g["grid"], "c": vals})
    return out


def main():
    work, tag, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
    trace, n_proofs = last_proof(trace_launches(os.path.join(work, "serial_trace")))
    notes = [f"kernel trace: {n_proofs} proofs in the run, the last untimed one taken: {len(trace)} launches"]
    passes = {}
    for name in ("insts", "valu", "fetch", "write", "cycles"):
        seq = pmc_launches(os.path.join(work, f"serial_{name}"))
        if not seq:
            notes.append(f"pass {name}: no output (the pass failed or a counter is unknown on this image)")
            continue
        proof, _ = last_proof(seq)
        same = len(proof) == len(trace) and all(a["name"] == b["name"] and a["grid"] == b["grid"] for a, b in zip(proof, trace))
        passes[name] = (proof, same)
        notes.append(f"pass {name}: {len(proof)} launches, sequence identical to the trace's: {same}")
    # the library's launch log of the traced run: the five commitment jobs of a proof, exactly
    log = None
    lp = os.path.join(work, "serial_trace_acclog.json")
    if os.path.exists(lp):
        launches = json.load(open(lp))["launches"]
        per_proof = sum(1 for d in trace if d["name"] == "sg::msm_accumulate")
        # the last untimed proof is the second to last of the run
        if per_proof and len(launches) >= 2 * per_proof:
            log = launches[-2 * per_proof:-per_proof]
    # ---- join launch by launch
    launches = []
    job = -1
    lincomb_i = 0
    lincomb_inputs = [0, 11, 6]    # lincomb_kernel launches of a proof: instance column, f, L (include/summa_prover.hpp; the rotation sets are lincomb_sets_kernel)
    for i, d in enumerate(trace):
        rec = dict(d)
        for pname, (proof, same) in passes.items():
            if same:
                rec.update(proof[i]["c"])
        if d["name"] == "sg::msm_digits":
            job += 1
        rec["job"] = job if d["name"].startswith("sg::msm_") else None
        if d["name"] == "sg::lincomb_kernel":
            rec["inputs"] = lincomb_inputs[lincomb_i] if lincomb_i < len(lincomb_inputs) else None
            lincomb_i += 1
        launches.append(rec)
    # passes whose sequence differs: per-name totals only
    by_name_extra = {}
    for pname, (proof, same) in passes.items():
        if not same:
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            cnt = collections.Counter()
            for d in proof:
                cnt[d["name"]] += 1
                for c, v in d["c"].items():
                    agg[d["name"]][c] += v
            by_name_extra[pname] = (agg, cnt)

    def job_rec(j):
        return log[j] if log and j is not None and 0 <= j < len(log) else None

    def min_bytes(rec):
        """(bytes, model) the launch must move as written: every input read once, every output written once"""
        nm, g, j = rec["name"], rec["grid"], job_rec(rec.get("job"))
        if nm.startswith("sg::msm_") and j is None:
            return None, None
        if j:
            M, n, E, L = j["M"], j["n"], j["entries"], j["task_len"]
            NB = M << 15                                  # fixed-base jobs at k = 17: one set of 2^15 buckets per polynomial
            T = E // L + min(NB, E) // 2                  # tasks: full ones + on average half a short one per non-empty bucket
            W = 16
        if nm == "sg::msm_accumulate":
            return E * (4 + 64) + T * 144, "entries x (4 B ref + 64 B point) + tasks x 144 B partial sums (SURVEY's algorithmic figure is 96 B per (scalar, point) pair of the job: `algorithmic_bytes`)"
        if nm == "sg::msm_digits":
            return M * n * (R + 2 * W), "M n x (32 B scalar + 16 x 2 B digits)"
        if nm == "sg::msm_hist":
            return M * n * 2 * W, "digits read once"
        if nm == "sg::msm_partition":
            return M * n * 2 * W + 6 * E, "digits in, 6 B (bucket, ref) pairs out"
        if nm == "sg::msm_fine_sort":
            return 6 * E + 4 * E, "pairs in, 4 B refs out"
        if nm == "sg::msm_fine_sort_fused":
            return 6 * E + 4 * E + 16 * NB, "pairs in, 4 B refs out, four words per bucket out (count, offset, tasks, task slot); entries = the job's nominal W x n"
        if nm == "sg::msm_task_scatter_reserve":
            return 8 * NB + 8 * T, "bucket counts in (two passes), task table out"
        if nm in ("sg::msm_hist_prefix", "sg::msm_scan_blocks", "sg::msm_scan_sums", "sg::msm_scan_write", "sg::msm_scan_small", "sg::msm_task_scan"):
            return 8 * NB, "bucket counters in and out"
        if nm == "sg::msm_task_hist":
            return 4 * NB, "bucket counts in"
        if nm == "sg::msm_task_scatter":
            return 4 * NB + 8 * T, "bucket counts in, task table out"
        if nm.startswith("sg::msm_fold_buckets"):
            return 144 * (T + NB), "partial sums in, one sum per bucket out"
        if nm.startswith("sg::msm_reduce2d_lines"):
            return 144 * NB, "bucket sums in (line sums out are 2^-7 of that)"
        if nm.startswith("sg::msm_reduce2d_bits"):
            return 144 * 2 * (1 << 8) * M, "line sums in"
        if nm == "sg::msm_export_points":
            return 3 * 96 * M * 16, "window sums to the host"
        if nm == "sg::ntt_pass":
            return 2 * g * 2 * R, "2 elements per thread x (32 B in + 32 B out); inter-pass twiddle tables (32 B per element of the passes that have one) not counted"
        if nm == "sg::ntt_pass_r4":   # two stages per sweep: four elements per thread
            return 4 * g * 2 * R, "4 elements per thread x (32 B in + 32 B out); inter-pass twiddle tables (32 B per element of the passes that have one) not counted"
        if nm.startswith("sg::numerator_fused_kernel"):
            return NE * R * (9 + 3 + 1 + 2 + 6 + 3 + 3 + 1), ("5n rows x (9 fixed + 3 advice + instance columns of the gate programs, 2 z, 6 sigma, 3 selector "
                                                               "columns, lz / a' / s', values out) x 32 B: every column once")
        if nm == "sg::lincomb_sets_kernel":
            return N * R * (31 + 5), "31 inputs + 5 outputs x n x 32 B"
        if nm == "sg::quot_perm_kernel":
            return NE * R * (2 + 6 + 6 + 3 + 1 + 1), "5n rows x (2 z + 6 columns + 6 sigma + 3 selector columns + values in + values out) x 32 B"
        if nm == "sg::quot_lookup_kernel":
            return NE * R * (1 + 2 + 2 + 3 + 1 + 1), "5n rows x (z, a', s', input, table, 3 selector columns, values in, values out) x 32 B"
        if nm.startswith("sg::gates_fixed_kernel"):
            return NE * R * (9 + 3 + 1), "5n rows x (9 fixed + 3 advice columns read, values out) x 32 B"
        if nm.startswith("sg::gates_kernel"):
            return g * R * 3, "rows x (two columns in, one out) x 32 B (the lookup's input expression)"
        if nm == "sg::coset_scale_kernel":
            cols = g // N
            return cols * N * R * 6 + NE * R, "per column n in + 5n out; the 5n-row power table once"
        if nm.startswith("sg::coset_combine_kernel"):
            return NE * R * 3, "5n values + 5n inverse powers in, 5 pieces of n out"
        if nm.startswith("sg::eval_poly_batch_kernel"):
            return g * R, "one coefficient per thread of the first level (40 polynomials x n x 32 B); the second level reads the block sums"
        if nm == "sg::lincomb_kernel":
            m = rec.get("inputs")
            return (N * R * (m + 1), "(inputs + 1 output) x n x 32 B") if m is not None else (None, None)
        if nm == "sg::batch_invert_kernel":
            return 3 * N * R * 2, "3 n denominators in and out"
        if nm == "sg::grand_fraction_kernel":
            return N * R * (6 + 6 + 4 + 3) // 1, "value, sigma and lookup columns in (16 n), three n-row outputs -- per pass (two passes: denominators, numerators)"
        if nm.startswith("sg::prefix_product_"):
            return 3 * N * R * 2 if "write" in nm else 3 * N * R, "three n-row columns (read; the write pass also writes them)"
        if nm.startswith("sg::kate_"):
            b = 11 if "batch" in nm else 1
            return (b * N * R * 2 if "write" in nm else b * N * R), "quotient columns (read; the write pass also writes them)"
        if nm == "sg::fr_random_kernel":
            return g * R, "one element per thread out"
        if nm == MARKER:
            return 3 * N * R, "three advice columns in"
        if nm == "sg::to_mont_kernel" or nm == "sg::scale_by_device_scalar_kernel":
            return g * R * 2, "one element per thread in and out"
        if nm.startswith("sg::lookup_permute_"):
            return 2 * N * R, "input and table columns (hist) / permuted columns (write)"
        return None, None

    for rec in launches:
        b, model = min_bytes(rec)
        rec["min_bytes"] = b
        rec["min_bytes_model"] = model
        j = job_rec(rec.get("job"))
        if rec["name"] == "sg::msm_accumulate" and j:
            rec["algorithmic_bytes"] = 96 * j["M"] * j["n"]
            rec["job_shape"] = {k_: j[k_] for k_ in ("M", "n", "entries", "threads", "task_len")}

    # ---- per kernel name
    names = collections.OrderedDict()
    for rec in launches:
        names.setdefault(rec["name"], []).append(rec)
    kernels = []
    counters = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU")
    for nm, recs in names.items():
        us = sum(r["us"] for r in recs)
        k = {"kernel": nm, "launches": len(recs), "us": round(us, 1)}
        for c in counters:
            if all(c in r for r in recs):
                k[c] = sum(r[c] for r in recs)
        for c in ("VALUBusy", "VALUUtilization"):
            if all(c in r for r in recs):
                k[c + "_pct"] = round(sum(r[c] * r["us"] for r in recs) / us, 2)      # time-weighted
        # FETCH_SIZE / WRITE_SIZE: KB as rocprofv3 reports them
        if all("FETCH_SIZE" in r for r in recs):
            k["fetch_bytes_counted"] = sum(r["FETCH_SIZE"] for r in recs) * 1024
        if all("WRITE_SIZE" in r for r in recs):
            k["write_bytes_counted"] = sum(r["WRITE_SIZE"] for r in recs) * 1024
        for pname, (agg, cnt) in by_name_extra.items():
            if nm in agg and cnt[nm] == len(recs):
                for c, v in agg[nm].items():
                    if c in ("VALUBusy", "VALUUtilization"):
                        k[c + "_pct"] = round(v / cnt[nm], 2)
                    elif c == "FETCH_SIZE":
                        k["fetch_bytes_counted"] = v * 1024
                    elif c == "WRITE_SIZE":
                        k["write_bytes_counted"] = v * 1024
                    else:
                        k[c] = v
        if all(r["min_bytes"] is not None for r in recs):
            k["min_bytes"] = sum(r["min_bytes"] for r in recs)
            k["min_bytes_model"] = recs[0]["min_bytes_model"]
            k["frac_hbm_min_bytes"] = round(k["min_bytes"] / (us * 1e-6) / HBM_PEAK, 4)
        if nm == "sg::msm_accumulate" and all("algorithmic_bytes" in r for r in recs):
            k["algorithmic_bytes"] = sum(r["algorithmic_bytes"] for r in recs)
            k["frac_hbm_algorithmic"] = round(k["algorithmic_bytes"] / (us * 1e-6) / HBM_PEAK, 4)
        if "fetch_bytes_counted" in k and "write_bytes_counted" in k:
            # the guide: FETCH_SIZE counts a wide coalesced streaming read at half its bytes on gfx950; 64-byte gathers read 0.94-1.00
            # (profiles/r04_fetch_calibration.json).  `stream` kernels read whole columns 16 B per lane and more: x2 applies to them
            gathers = nm in ("sg::msm_accumulate",)
            k["fetch_correction"] = "none (64-byte gathers: calibrated 0.94-1.00)" if gathers else "x2 (wide coalesced streams, guide)"
            k["traffic_bytes"] = k["fetch_bytes_counted"] * (1 if gathers else 2) + k["write_bytes_counted"]
            k["frac_hbm_traffic"] = round(k["traffic_bytes"] / (us * 1e-6) / HBM_PEAK, 4)
            if k.get("min_bytes"):
                k["traffic_over_min_bytes"] = round(k["traffic_bytes"] / k["min_bytes"], 2)
        if "VALUBusy_pct" in k:
            k["issue_floor_us"] = round(us * k["VALUBusy_pct"] / 100.0, 1)     # the time its vector ALUs were issuing: its own floor at this instruction mix
        if "SQ_WAVE_CYCLES" in k and k["SQ_WAVE_CYCLES"]:
            wc = k["SQ_WAVE_CYCLES"]
            k["wave_cycles_split_pct"] = {"parked (s_waitcnt / barrier)": round(100 * k.get("SQ_WAIT_ANY", 0) / wc, 1),
                                          "issue-stalled": round(100 * k.get("SQ_WAIT_INST_ANY", 0) / wc, 1),
                                          "issuing": round(100 * k.get("SQ_ACTIVE_INST_ANY", 0) / wc, 1)}
        if "SQ_INSTS_VALU" in k and "VALUBusy_pct" in k and k["SQ_INSTS_VALU"]:
            # clocks per VALU wave-instruction per SIMD while busy, at the nominal 2.4 GHz (the effective clock is lower under load)
            k["busy_clk_per_valu_inst_at_2p4GHz"] = round(k["issue_floor_us"] * 1e-6 * 2.4e9 / (k["SQ_INSTS_VALU"] / SIMDS), 2)
        bound = None
        if "VALUBusy_pct" in k:
            hb = k.get("frac_hbm_traffic", 0) or 0
            bound = "valu" if k["VALUBusy_pct"] >= 60 else ("hbm" if hb >= 0.35 else "latency / occupancy")
        k["bound"] = bound
        kernels.append(k)
    kernels.sort(key=lambda k: -k["us"])
    total_us = sum(k["us"] for k in kernels)
    floor_us = sum(k.get("issue_floor_us", 0) for k in kernels)
    insts = sum(k.get("SQ_INSTS_VALU", 0) for k in kernels)
    budget = {
        "tag": tag, "what": "one k = 17 MstInclusion proof (LEVELS 20, N_CURRENCIES 2), compiled driver, SG_PROVER_SERIAL=1: every kernel alone on the device",
        "reference_region": "/root/reference/zk_prover/src/circuits/utils.rs:88-105 (the reference's only timer wraps create_proof)",
        "notes": notes,
        "launches": len(launches), "kernel_us_total": round(total_us, 1),
        "valu_wave_instructions_total": insts,
        "issue_floor_ms": round(floor_us / 1e3, 3),
        "issue_floor_definition": "sum over kernels of duration x VALUBusy: the time the vector ALUs were issuing -- what is left if every stall, tail and launch gap went away at the present instruction counts",
        "mean_busy_clk_per_valu_inst_at_2p4GHz": round(floor_us * 1e-6 * 2.4e9 / (insts / SIMDS), 2) if insts else None,
        "kernel_time_over_issue_floor": round(total_us / floor_us, 3) if floor_us else None,
        "kernels": kernels,
        "msm_jobs": log,
    }
    json.dump(budget, open(out_path, "w"), indent=1)
    print(f"# {tag}: {len(launches)} launches, {total_us / 1e3:.3f} ms of kernel time, issue floor {floor_us / 1e3:.3f} ms, {insts / 1e6:.0f} M VALU wave-instructions")
    print(f"# {'kernel':44} {'n':>3} {'us':>8} {'VALU M':>8} {'busy%':>6} {'lanes%':>6} {'fetch MB':>9} {'write MB':>9} {'min MB':>8} {'hbm%':>6} {'bound'}")
    for k in kernels:
        f = lambda key, sc=1.0, fmt="{:8.1f}": (fmt.format(k[key] * sc) if key in k and k[key] is not None else " " * 7 + "-")
        print(f"  {k['kernel'][:44]:44} {k['launches']:3d} {k['us']:8.1f} {f('SQ_INSTS_VALU', 1e-6)} {f('VALUBusy_pct', 1, '{:6.1f}')} {f('VALUUtilization_pct', 1, '{:6.1f}')} "
              f"{f('fetch_bytes_counted', 1e-6, '{:9.1f}')} {f('write_bytes_counted', 1e-6, '{:9.1f}')} {f('min_bytes', 1e-6)} {f('frac_hbm_traffic', 100, '{:6.1f}')} {k['bound']}")
    for n_ in notes:
        print("#", n_)


if __name__ == "__main__":
    main()
