"""The measurement tooling of round 5 on synthetic rocprofv3 databases (no GPU, no profiler): tools/proof_budget.py joins a kernel
trace and separate --pmc passes launch by launch and prices the issue floor; tools/batch_budget.py takes per-proof instruction
counts from the difference of two passes; tools/summarize_prof.py attributes msm_accumulate launches through the library's launch
log.  The databases have the views the tools read (`kernels`, `counters_collection`) with the columns rocprofv3 7.x gives them."""
import json
import os
import sqlite3
import subprocess
import sys

from conftest import ROOT

PROOF = ["sg::count_noncanonical_kernel(x)", "sg::fr_random_kernel(a)", "sg::msm_digits(a)", "sg::msm_hist()", "sg::msm_accumulate(a,b)",
         "void sg::msm_fold_buckets<1>(a)", "sg::ntt_pass(sg::PassArgs)", "sg::lincomb_kernel(a)", "sg::msm_digits(a)", "sg::msm_accumulate(a,b)",
         "sg::quot_perm_kernel(a)"]
US = {"sg::msm_accumulate": 300.0, "sg::ntt_pass": 80.0}


def _trace_db(path, proofs, grid=131072):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    db = sqlite3.connect(path)
    db.execute("create table kernels (name, start, end, duration, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, lds_size, scratch_size, queue_id)")
    t = 0
    for _ in range(proofs):
        for nm in PROOF:
            dur = int(1000 * US.get(nm.split("(")[0], 10.0))
            db.execute("insert into kernels values (?,?,?,?,?,?,?,?,?,?,?,?)", (nm, t, t + dur, dur, grid, 1, 1, 256, 64, 0, 0, 1))
            t += dur + 500
    db.commit()


def _pmc_db(path, proofs, counters, grid=131072):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    db = sqlite3.connect(path)
    db.execute("create table counters_collection (kernel_name, grid_size, queue_id, counter_name, value, start, dispatch_id)")
    t, did = 0, 0
    for _ in range(proofs):
        for nm in PROOF:
            did += 1
            for c, v in counters.items():
                val = v(nm) if callable(v) else v
                db.execute("insert into counters_collection values (?,?,?,?,?,?,?)", (nm, grid, 1, c, val, t, did))
            t += 6000
    db.commit()


def test_proof_budget_joins_the_passes_launch_by_launch(tmp_path):
    work = tmp_path / "work"
    _trace_db(str(work / "serial_trace" / "h" / "1_results.db"), 6)
    _pmc_db(str(work / "serial_insts" / "h" / "1_results.db"), 4, {"SQ_INSTS_VALU": lambda n: 4.0e6 if "accumulate" in n else 1.0e5, "SQ_WAVES": 2048.0})
    _pmc_db(str(work / "serial_valu" / "h" / "1_results.db"), 4, {"VALUBusy": lambda n: 60.0 if "accumulate" in n else 20.0, "VALUUtilization": 99.0})
    _pmc_db(str(work / "serial_fetch" / "h" / "1_results.db"), 4, {"FETCH_SIZE": 1000.0})
    _pmc_db(str(work / "serial_write" / "h" / "1_results.db"), 4, {"WRITE_SIZE": 500.0})
    log = [dict(entries=6 << 21, n=1 << 17, M=6, threads=131072, fixed=1, jobs_in_flight=1, task_len=16),
           dict(entries=1 << 21, n=1 << 17, M=1, threads=131072, fixed=1, jobs_in_flight=1, task_len=16)] * 6
    (work / "serial_trace_acclog.json").write_text(json.dumps({"launches": log}))
    out = tmp_path / "budget.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "proof_budget.py"), str(work), "t", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads(out.read_text())
    assert b["launches"] == len(PROOF) and all("sequence identical to the trace's: True" in n for n in b["notes"] if n.startswith("pass ") and "no output" not in n)
    assert any("pass cycles: no output" in n for n in b["notes"])           # a pass that is missing is said to be missing
    acc = next(k for k in b["kernels"] if k["kernel"] == "sg::msm_accumulate")
    assert acc["launches"] == 2 and acc["us"] == 600.0 and acc["SQ_INSTS_VALU"] == 8.0e6 and acc["VALUBusy_pct"] == 60.0
    assert acc["issue_floor_us"] == 360.0 and acc["bound"] == "valu"
    assert acc["algorithmic_bytes"] == 96 * (6 + 1) * (1 << 17)                 # the two jobs of the launch log, 96 B per (scalar, point)
    assert acc["fetch_correction"].startswith("none") and acc["traffic_bytes"] == (2 * 1000 + 2 * 500) * 1024
    ntt = next(k for k in b["kernels"] if k["kernel"] == "sg::ntt_pass")
    assert ntt["traffic_bytes"] == (2 * 1000 + 500) * 1024 and ntt["min_bytes"] == 2 * 131072 * 64        # x2 on FETCH for a wide stream
    floor = sum(k.get("issue_floor_us", 0) for k in b["kernels"])
    assert abs(b["issue_floor_ms"] * 1e3 - floor) < 0.5 and abs(b["kernel_time_over_issue_floor"] - b["kernel_us_total"] / floor) < 1e-2
    # ---- the batch: per-proof instructions from the difference of two passes, priced with the serial proof's busy time
    insts = {"SQ_INSTS_VALU": lambda n: 4.0e6 if "accumulate" in n else 1.0e5, "SQ_WAVES": 2048.0}
    _pmc_db(str(tmp_path / "a" / "h" / "1_results.db"), 10, insts)
    _pmc_db(str(tmp_path / "b" / "h" / "1_results.db"), 26, insts)
    (tmp_path / "a.json").write_text(json.dumps({"proofs_made_in_run": 10}))
    (tmp_path / "b.json").write_text(json.dumps({"proofs_made_in_run": 26}))
    (tmp_path / "line.json").write_text(json.dumps({"proofs_per_s": 1000.0}))
    out2 = tmp_path / "batch.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "batch_budget.py"), str(tmp_path / "a"), str(tmp_path / "b"), str(tmp_path / "a.json"),
                        str(tmp_path / "b.json"), str(out), str(tmp_path / "line.json"), str(out2)], capture_output=True, text=True, cwd=os.path.join(ROOT, "tools"))
    assert r.returncode == 0, r.stderr[-2000:]
    bb = json.loads(out2.read_text())
    assert bb["valu_wave_instructions_per_proof"] == 8.0e6 + 9 * 1.0e5 and bb["launches_per_proof"] == len(PROOF)
    assert abs(bb["issue_floor_ms_per_proof"] - b["issue_floor_ms"]) < 1e-3          # the same proof, so the same floor
    assert abs(bb["efficiency_issue_floor_over_time_per_proof"] - b["issue_floor_ms"]) < 1e-3   # at 1000 proofs/s: floor [ms] / 1 ms


def test_summarize_prof_attributes_accumulations_through_the_launch_log(tmp_path):
    """two jobs share a grid (the launch is sized by an upper bound): the launch log tells them apart, record i = i-th launch"""
    repo = tmp_path / "repo"
    (repo / "gpurun_out").mkdir(parents=True)
    (repo / "profiles").mkdir()
    _trace_db(str(repo / "gpurun_out" / "prof_tt" / "h" / "1_results.db"), 3)
    log = [dict(entries=1 << 24, n=1 << 20, M=1, threads=131072, fixed=0, jobs_in_flight=3, task_len=64),
           dict(entries=1 << 21, n=1 << 17, M=1, threads=131072, fixed=1, jobs_in_flight=1, task_len=16)] * 3
    (repo / "gpurun_out" / "prof_tt_acclog.json").write_text(json.dumps({"launches": log}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_prof.py"), "tt"], capture_output=True, text=True, cwd=str(repo))
    assert r.returncode == 0, r.stderr[-2000:]
    s = json.loads((repo / "profiles" / "tt_summary.json").read_text())
    assert s["msm_accumulate_attribution"]["kernel_trace"]["method"].startswith("library launch log")
    jobs = {(j["job_threads"], j["jobs_in_flight_at_issue"]): j["launches"] for j in s["msm_accumulate_by_job"]}
    assert jobs == {(1 << 20, 3): 3, (1 << 17, 1): 3}
    # a log that ends before the trace does (written before the run's last launches) attributes the launches it holds
    (repo / "gpurun_out" / "prof_tt_acclog.json").write_text(json.dumps({"launches": log[:-1]}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_prof.py"), "tt"], capture_output=True, text=True, cwd=str(repo))
    assert r.returncode == 0, r.stderr[-2000:]
    s = json.loads((repo / "profiles" / "tt_summary.json").read_text())
    assert s["msm_accumulate_attribution"]["kernel_trace"]["launches_beyond_the_log"] == 1
    # a log with MORE records than the trace has launches is another run's: not used
    (repo / "gpurun_out" / "prof_tt_acclog.json").write_text(json.dumps({"launches": log + log}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_prof.py"), "tt"], capture_output=True, text=True, cwd=str(repo))
    assert r.returncode == 0, r.stderr[-2000:]
    s = json.loads((repo / "profiles" / "tt_summary.json").read_text())
    assert "heuristic" in s["msm_accumulate_attribution"]["kernel_trace"]["method"]
