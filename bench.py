#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): MSM-points/sec on the standalone BN254 G1 MSM of 2^20
uniform scalars x synthetic-SRS points (configs[1]); one "step" = one such MSM per GPU with
inputs resident in HBM.  With --gpus N (launched by torch.distributed.run, one rank per GPU)
the N shards form one N*2^20-point MSM: every rank reduces its shard, the 64-byte partial
points are exchanged with one RCCL all_gather and summed (weak scaling, per-GPU work fixed).

`python bench.py --gpus N` launched bare starts its own N ranks (a torch.distributed.run child, before any GPU call).

Also reported in the same JSON line: the NTT rates (configs[2]), the MSM+NTT kernel sum for
the k=17 proof op list, a real and VERIFIED k=17 proof through the reference's API
(configs[3]) with the restated-reference CPU op list beside it, the proof-level batch (configs[4] in small:
`batch_k17`, `proofs_per_s`), `roofline` for the dominant kernel (msm_accumulate) and `cpu_baseline` (the oracle's
restated halo2 best_multiexp on this box's host cores)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MSM_BYTES_PER_PAIR = 96        # 64 B point + 32 B scalar
NTT_BYTES_PER_ELEM = 64        # 32 B read + 32 B write, one logical pass
MADS_PER_MIXED_ADD = 6 * 162 + 2 * 126 + 243   # v_mad_u64_u32 per XYZZ += affine
MAD_PEAK = 30.1e12             # measured v_mad_u64_u32 lane-ops/s, full occupancy (microbench2)


def pmc_traffic(kernel, log_n, grid_threads=None, job_threads=None):
    """HBM-side bytes per launch of `kernel` from the newest committed PMC summary
    (profiles/*_summary.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    same command at the default 2^20 workload); null when no matching profile exists."""
    import glob
    if log_n != LOG_N:
        return {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")), reverse=True):
        pm = json.load(open(path)).get("pmc", {})
        exact = pm.get(f"{kernel}@grid{grid_threads}@job{job_threads}")   # the timed job's own launches (same grid AND job shape)
        if exact and "FETCH_SIZE_KB_avg" in exact and "WRITE_SIZE_KB_avg" in exact:
            pm = {f"{kernel}@grid{grid_threads}": exact}
        cands = [(int(k.split("@grid")[1]), v) for k, v in pm.items()
                 if k.startswith(kernel + "@") and "@job" not in k and "FETCH_SIZE_KB_avg" in v and "WRITE_SIZE_KB_avg" in v]
        if cands:
            # the same kernel also runs at other sizes (fused batches): take the launch shape of the timed MSM
            _, v = min(cands, key=lambda gv: abs(gv[0] - grid_threads)) if grid_threads else max(cands)
            raw = (v["FETCH_SIZE_KB_avg"] + v["WRITE_SIZE_KB_avg"]) * 1024
            # the guide's x2 correction of FETCH_SIZE holds for wide coalesced streams; this kernel's reads are 64-byte gathers
            # at unrelated indices, calibrated on a known byte count of exactly that pattern (tools/calib_gather.hip)
            cal, cal_src = None, None
            for cpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_fetch_calibration.json")), reverse=True):
                rows = json.load(open(cpath))
                cal = {f"table_{r['table_MiB']}MiB": round(r["fetch_over_requested"], 3) for r in rows}
                cal_src = os.path.basename(cpath)
                break
            return {"traffic": raw, "traffic_source": os.path.basename(path), "fetch_size_over_requested_bytes_for_64B_gathers": cal,
                    "traffic_note": "FETCH_SIZE + WRITE_SIZE of the launch (separate --pmc passes).  No x2 correction: for this kernel's access "
                                    f"pattern, 64-byte gathers, FETCH_SIZE reads 0.94-1.00 x the bytes requested ({cal_src}: a 64 MiB table "
                                    "resident in the Infinity Cache, as the MSM's base table is, and a 1 GiB table); every point is gathered once per "
                                    "window, hence ~14 x the algorithmic bytes, served mostly on-die"}
    return {}


def snapshot_tree(levels=20, nc=2):
    """a synthetic snapshot of 2^levels users (the reference bench downloads its CSVs from S3): username
    field elements from ChaCha20, 40-bit balances; the Merkle sum tree is built and kept on the device"""
    import torch
    from circuits_halo2_amd import arithmetic as A
    from circuits_halo2_amd.merkle_sum_tree import DeviceMerkleSumTree
    from circuits_halo2_amd.utils import random_fr_canonical
    size = 1 << levels
    d_users = A.fr_random(bytes(range(32)), 1, size)
    bal = random_fr_canonical(77, size * nc).reshape(-1, 32).copy()
    bal[:, 5:] = 0                                           # sums stay below 2^64 (N_BYTES = 8)
    d_bals = A.fr_to_montgomery(torch.from_numpy(bal.reshape(-1)).cuda())
    return DeviceMerkleSumTree(d_users, d_bals, levels, nc)


def oracle_vk(params, vk):
    """the verifying key in the form the oracle's restated verifier takes (checker leg only)"""
    from oracle import pyref as PR
    f2 = lambda b: (PR.fq_from_bytes(b[:32]), PR.fq_from_bytes(b[32:64]))
    s_g2 = (f2(params.s_g2[:64]), f2(params.s_g2[64:]))
    return {"k": vk.k, "n_currencies": vk.n_currencies, "vk_digest": vk.transcript_repr, "fixed_comms": vk.fixed_comms,
            "permutation_comms": vk.permutation_comms, "g2": (f2(params.g2[:64]), f2(params.g2[64:])),
            "neg_s_g2": (s_g2[0], ((-s_g2[1][0]) % PR.Q, (-s_g2[1][1]) % PR.Q))}


class FailSafe:
    """ONE JSON line on stdout whatever happens to this run (rank 0 prints; every rank leaves).

    * `beat(label, allow_s)`: the main thread announces a step that may block (a collective, a barrier, a batch) and how
      long it may take; a watchdog thread fires when the step overstays, or when the whole run passes `limit_s`.  The
      allowances are shorter than the process group's own timeout (90 s): with the nccl backend that timeout ends the
      process from a C++ thread, with no chance to print anything.
    * SIGTERM / SIGINT (what torch.distributed.run sends the surviving ranks when one rank has died) reach the watchdog
      through the signal module's wake-up descriptor, which is written at C level even while the main thread sits in a
      collective.
    * On any of these -- and on an exception out of main() -- rank 0 prints {"metric", "value": null, "error", "partial"}
      with whatever had been measured, and the process leaves with os._exit (no destructors: a hung collective would hang
      them too)."""

    def __init__(self, rank, args, limit_s):
        import threading
        self.rank, self.args, self.limit_s = rank, args, limit_s
        self.partial = {}
        self.lock = threading.Lock()
        self.printed = False
        self.t0 = time.monotonic()
        self.step_label, self.step_deadline = "start", self.t0 + limit_s
        self.rfd, self.wfd = os.pipe()
        os.set_blocking(self.wfd, False)
        self.thread = None
        self.base_line = None      # the headline part of the line, whole and valid, once the timed steps are done (rank 0)

    def arm(self):
        import signal
        import threading
        try:   # (only possible in the main thread of the main interpreter; a test harness importing main() elsewhere goes without)
            for sig in (signal.SIGTERM, signal.SIGINT):
                signal.signal(sig, lambda *_: None)
            signal.set_wakeup_fd(self.wfd, warn_on_full_buffer=False)
        except ValueError:
            pass
        self.thread = threading.Thread(target=self._watch, name="bench-failsafe", daemon=True)
        self.thread.start()

    def beat(self, label, allow_s):
        self.step_label, self.step_deadline = label, time.monotonic() + allow_s

    def _watch(self):
        import select
        while True:
            now = time.monotonic()
            wait = max(0.05, min(self.step_deadline, self.t0 + self.limit_s) - now)
            ready, _, _ = select.select([self.rfd], [], [], min(wait, 1.0))
            if ready:
                sig = os.read(self.rfd, 16)
                self.fail(f"signal {list(sig)} while in step '{self.step_label}' (another rank failed, or the launcher gave up)", 143)
            now = time.monotonic()
            if now > self.t0 + self.limit_s:
                self.out_of_time(f"wall-clock limit of {self.limit_s:.0f} s reached in step '{self.step_label}'")
            if now > self.step_deadline:
                self.out_of_time(f"step '{self.step_label}' overstayed its allowance (a rank is missing from a collective?)")

    def error_line(self, reason):
        a = self.args
        return {"metric": "msm_points_per_sec", "value": None, "unit": "points/s", "n_gpus": a.gpus, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u32", "data": "synthetic", "error": reason, "elapsed_s": round(time.monotonic() - self.t0, 1),
                "partial": self.partial}

    def emit(self, line):
        """the run's one line (rank 0); False if the watchdog got there first"""
        with self.lock:
            if self.printed:
                return False
            self.printed = True
            print(json.dumps(line), flush=True)
            return True

    def extra_failed(self, reason):
        """an EXTRA threw after the headline was measured (single-rank run): the headline is printed as the run's line with the
        failure named in `extras_failed`, and the run exits NON-ZERO -- a wrong result in an extra is not a time-out"""
        line = dict(self.base_line, extras_failed=reason, partial_extras=self.partial)
        self.emit(line)
        os._exit(3)

    def out_of_time(self, reason):
        """time ran out.  In a single-rank run whose timed steps are done, what overstayed is an EXTRA: the headline was
        measured in full and is printed as the run's line, the missing extras named in `incomplete` (exit code 0).  In a
        multi-rank run a step that overstays means a rank is gone: the measurement is void, the error line says so."""
        if self.args.gpus == 1 and self.base_line is not None:
            line = dict(self.base_line, incomplete=reason, partial_extras=self.partial)
            self.emit(line)
            os._exit(0)
        self.fail(reason, 124)

    def fail(self, reason, code):
        if self.rank == 0:
            self.emit(self.error_line(reason))
        else:
            print(f"bench.py rank {self.rank}: {reason}", file=sys.stderr, flush=True)
        os._exit(code)


_FS = None      # the run's FailSafe (set by main)


def _beat(label, allow_s=75.0):
    if _FS is not None:
        _FS.beat(label, allow_s)


def run_ranks_and_relay(args, argv):
    """`python bench.py --gpus N` launched bare: start the N ranks (one torch.distributed.run child, its own process
    group), relay rank 0's JSON line, and keep the promise of FailSafe from the outside too: when the child fails,
    prints nothing, or outlives `--wall-limit` + 30 s, its whole process group is killed (by the group id this parent
    created -- never by pattern) and the parent prints the error line itself.  Returns the exit code."""
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.monotonic()
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    lines = []

    def pump():
        for ln in child.stdout:
            if ln.startswith("{"):
                lines.append(ln.strip())
            else:
                sys.stdout.write(ln)
                sys.stdout.flush()
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    reason = None
    try:
        rc = child.wait(timeout=args.wall_limit + 30)
    except subprocess.TimeoutExpired:
        reason, rc = f"the ranks outlived --wall-limit {args.wall_limit:.0f} s + 30 s", 124
    if rc != 0 and reason is None:
        reason = f"torch.distributed.run exited with code {rc} (a rank died or failed)"
    if reason is not None:
        try:
            os.killpg(child.pid, signal.SIGKILL)     # child.pid is the id of the session / group created above
        except ProcessLookupError:
            pass
        child.wait()
    th.join(timeout=5)
    if lines and reason is None:
        print(lines[-1], flush=True)
        return 0
    rank0 = None
    if lines:
        try:
            rank0 = json.loads(lines[-1])
        except ValueError:
            rank0 = {"unparsed": lines[-1][:400]}
    if rank0 is not None and rank0.get("error"):
        print(json.dumps(rank0), flush=True)          # rank 0 already said what happened, with its partial results
        return rc or 1
    fs = FailSafe(0, args, args.wall_limit)
    fs.t0 = t0
    line = fs.error_line(reason or "the ranks printed no JSON line")
    if rank0 is not None:
        line["rank0_line"] = rank0
    print(json.dumps(line), flush=True)
    return rc or 1


def _all_max(x, world, coll_dev):
    """max over ranks of a float (every rank calls it)"""
    import torch
    import torch.distributed as dist
    if world == 1:
        return float(x)
    _beat("all_reduce(max)")
    t = torch.tensor([float(x)], device=coll_dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _all_sum(xs, world, coll_dev):
    import torch
    import torch.distributed as dist
    if world == 1:
        return [float(v) for v in xs]
    _beat("all_reduce(sum)")
    t = torch.tensor([float(v) for v in xs], device=coll_dev, dtype=torch.float64)
    dist.all_reduce(t)
    return [float(v) for v in t.tolist()]


# algorithmic bytes of ONE k = 17 proof's op list: 16 MSMs of 2^17 pairs at 96 B, 11 665 408
# element-transforms (9 iNTT 2^17 + 9 NTT 2^20 + 1 iNTT 2^20, halo2's extended-domain layout) at 64 B, and the quotient's
# numerator reading about 29 extended columns of 2^20 rows at 32 B
PROOF_ALG_BYTES = 16 * (1 << 17) * MSM_BYTES_PER_PAIR + 11_665_408 * NTT_BYTES_PER_ELEM + 29 * (1 << 20) * 32


def proof_roofline(ms, what):
    achieved = PROOF_ALG_BYTES / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "algorithmic_bytes": PROOF_ALG_BYTES, "ms": ms, "traffic": None,
            "note": what + "; algorithmic bytes of halo2's op list for this proof (16 x 2^17 x 96 B + 11 665 408 x 64 B + 29 x 2^20 x 32 B); "
                           "the kernels behind it are integer-VALU-bound (see `alu`), the fraction says how far the whole proof is from a pure stream"}


def committed_proof_budget():
    """the counter-backed budget of one k = 17 proof from the newest committed profile (profiles/*_proof_budget.json, written by
    tools/prof_proof_r05.sh + tools/proof_budget.py on the GPU box): per kernel -- launches, microseconds, VALU wave-instructions,
    VALU busy, counted FETCH / WRITE bytes, the bytes the launch must move, fractions of the HBM roofline and of its own issue floor --
    and the proof's issue floor; None when no profile is committed"""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_proof_budget.json")))
    if not paths:
        return None
    b = json.load(open(paths[-1]))
    keep = ("kernel", "launches", "us", "SQ_INSTS_VALU", "VALUBusy_pct", "VALUUtilization_pct", "fetch_bytes_counted", "write_bytes_counted",
            "fetch_correction", "traffic_bytes", "min_bytes", "algorithmic_bytes", "frac_hbm_traffic", "frac_hbm_min_bytes", "issue_floor_us",
            "wave_cycles_split_pct", "bound")
    return {"source": os.path.basename(paths[-1]), "what": b.get("what"), "launches": b.get("launches"), "kernel_us_total": b.get("kernel_us_total"),
            "valu_wave_instructions_total": b.get("valu_wave_instructions_total"), "issue_floor_ms": b.get("issue_floor_ms"),
            "issue_floor_definition": b.get("issue_floor_definition"), "kernel_time_over_issue_floor": b.get("kernel_time_over_issue_floor"),
            "kernels": [{k_: k[k_] for k_ in keep if k_ in k} for k in b.get("kernels", []) if k.get("us", 0) >= 20.0]}


def committed_batch_budget():
    """the batch's instruction budget per proof from the newest committed profile (profiles/*_batch_budget.json, tools/prof_batch_r05.sh)"""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_batch_budget.json")))
    if not paths:
        return None
    b = json.load(open(paths[-1]))
    out = {k_: b.get(k_) for k_ in ("what", "valu_wave_instructions_per_proof", "launches_per_proof", "issue_floor_ms_per_proof", "proofs_per_s",
                                     "ms_per_proof", "efficiency_issue_floor_over_time_per_proof")}
    out["source"] = os.path.basename(paths[-1])
    return out


def host_pointer_extra(scal, bases, want):
    """sg_msm_g1 / sg_commit / sg_ntt_fr at 2^20 with every argument in pageable host memory (numpy), timed around the C call
    itself: the link is inside the figure (96 MiB, 32 MiB, 32 MiB each way)"""
    import torch
    import circuits_halo2_amd as sg
    from circuits_halo2_amd import ffi
    L = sg.lib()
    n = 1 << 20
    hs, hb = scal[:32 * n].cpu().numpy().copy(), bases[:64 * n].cpu().numpy().copy()
    res = np.zeros(64, dtype=np.uint8)
    p_s, p_b, p_r = (C.c_void_p(x.ctypes.data) for x in (hs, hb, res))

    def best(fn, reps=6):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts)
    out = {"log_n": 20, "memory": "pageable (numpy)"}
    out["sg_msm_g1_ms"] = best(lambda: ffi.check(L.sg_msm_g1(p_s, p_b, C.c_size_t(n), p_r)))
    out["sg_msm_g1_matches"] = bool((res == want).all()) if scal.numel() == 32 * n else None
    params = sg.ParamsKZG(20, hb, hb)
    try:
        h = C.c_uint64(params.handle())
        out["sg_commit_ms"] = best(lambda: ffi.check(L.sg_commit(h, C.c_int(0), p_s, C.c_size_t(n), p_r)))
        out["sg_commit_matches"] = bool((res == want).all()) if scal.numel() == 32 * n else None
    finally:
        params.free()
    w = ffi.u8(sg.EvaluationDomain(2, 20).get_omega())
    ha = hs.copy()
    p_a = C.c_void_p(ha.ctypes.data)
    out["sg_ntt_fr_ms"] = best(lambda: ffi.check(L.sg_ntt_fr(p_a, ffi.ptr(w), C.c_uint32(20))))
    tp, dev = torch.from_numpy(hb), torch.empty(64 * n, dtype=torch.uint8, device="cuda")

    def h2d():
        dev.copy_(tp, non_blocking=True)
        torch.cuda.synchronize()
    out["h2d_pageable_GBs"] = 64 * n / best(h2d) / 1e6
    out["note"] = ("PCIe-inclusive wall clock of the C call, best of 6, inputs in pageable host memory: sg_msm_g1 moves 96 MiB and sg_commit 32 MiB "
                   "(against the resident SRS), each as two chunk jobs on two engines with the second chunk's upload under the first chunk's MSM; sg_ntt_fr 32 MiB each way; "
                   "h2d_pageable_GBs is the plain copy rate of this box (page-locked memory is no faster here, so there is no staging layer to add)")
    return out


def host_cpu_info():
    """what the process may use of the host: the affinity mask, and the CPU quota of its control group when one is set
    (cgroup v2 `cpu.max`, v1 `cfs_quota_us`): on the boxes of this pool the mask is the whole machine, the quota is not"""
    info = {"affinity_cores": len(os.sched_getaffinity(0)), "nproc": os.cpu_count(), "cgroup_cpu_quota_cores": None}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            info["cgroup_cpu_quota_cores"] = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                info["cgroup_cpu_quota_cores"] = q / per
        except (OSError, ValueError):
            pass
    return info


def cpu_share_rehearsal(args, in_flight, busy_cores):
    """The 1024-proof batch again with this process confined to a fraction of the host's cores (`--cpu-share`, one child
    process per setting: the affinity has to be in place before the HIP runtime starts its threads): what ONE rank gets
    when 8 / 4 / 2 ranks share the host.  The shares are fractions of what this process can actually use -- the control
    group's CPU quota when there is one, else the affinity mask.  Each child picks its own in-flight setting among
    {in_flight / 2, in_flight} and runs the whole batch once."""
    import subprocess
    info = host_cpu_info()
    usable = info["cgroup_cpu_quota_cores"] or info["affinity_cores"]
    out = {"usable_cores": usable, "full_share_cores_busy": busy_cores, "by_share": {}}
    for div in (8, 4, 2):
        cores = max(1, int(usable // div))
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--batch-only", "--cpu-share", str(cores), "--batch-proofs",
               str(args.batch_proofs), "--batch-repeats", "1", "--no-cpu", "--batch-in-flight", str(in_flight), "--wall-limit", "150"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
            got = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            out["by_share"][f"1/{div}"] = {"cores": cores, "proofs_per_s": got.get("proofs_per_s"), "host_cpu_ms_per_proof": got.get("host_cpu_ms_per_proof"),
                                           "host_cores_busy": got.get("host_cores_busy_per_gpu"), "in_flight": got.get("in_flight"),
                                           "errors": got.get("errors"), "error": got.get("error")}
        except Exception as ex:      # an extra: never costs the line
            out["by_share"][f"1/{div}"] = {"cores": cores, "error": repr(ex)}
    return out


def batch_extra(args, rank, world, coll_dev):
    """BASELINE configs[4] at its stated size: inclusion proofs for `--batch-proofs` users IN TOTAL (default 1024) of a
    2^20-user snapshot at k = 17 (MstInclusionCircuit<20,2,8>), through circuits_halo2_amd.batch: the setup artifacts are
    generated on rank 0 and broadcast (GPU to GPU with nccl), the users are dealt round-robin over the ranks (1024 / N
    each), several proofs in flight per GPU, every proof re-verified before it counts (create_proof_checked).  Every
    rank takes part in every collective whatever happens locally (a local failure is carried as a flag, never as a
    missing participant).  Returns the extra key's dict on rank 0 and what the single-proof extras reuse."""
    import torch
    import torch.distributed as dist
    from circuits_halo2_amd import batch as B
    levels, k, nc = 20, 17, 2
    total = args.batch_proofs
    t0 = time.perf_counter()
    _beat("setup artifacts: keygen on rank 0 + broadcast", 150.0)
    params, pk, vk = B.setup_on_all_ranks(k, None, levels, nc)
    setup_s = time.perf_counter() - t0
    _beat("snapshot tree", 120.0)
    params.precompute()
    tree = snapshot_tree(levels, nc)
    torch.cuda.synchronize()
    out = {"k": k, "levels": levels, "n_currencies": nc, "proofs_total": total, "proofs_per_gpu": total / world, "n_gpus": world,
           "setup_artifacts_s": setup_s}
    users = [(7919 * i + 13) % (1 << levels) for i in range(total)]

    import resource

    def cpu_seconds():
        """user + system CPU time of this process so far, all threads (the host side of the proofs: witness hand-over,
        transcripts, Fiat-Shamir scalars, five MSM tails and the pairing check of the re-verification per proof)"""
        ru = resource.getrusage(resource.RUSAGE_SELF)
        return ru.ru_utime + ru.ru_stime

    made = [0]      # proofs made by this run so far, warm-up and pre-sweep included (what a profile of the whole process contains)
    fault = os.environ.get("SUMMA_BENCH_FAULT", "")      # "rank:batch:seconds" -- tests/test_gpu_batch.py kills one rank mid-batch

    def timed_batch(subset, in_flight, headline=False):
        """one timed batch over `subset` (dealt over the ranks): (proofs, errors, seconds max-over-ranks, this rank's
        result, CPU seconds of all ranks)"""
        _beat("proof batch", 75.0 + 0.05 * len(subset))
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if headline and fault.startswith(f"{rank}:batch:"):
            import signal
            import threading
            threading.Timer(float(fault.split(":")[2]), lambda: os.kill(os.getpid(), signal.SIGKILL)).start()
        t1, c1 = time.perf_counter(), cpu_seconds()
        res, failure = None, 0
        try:
            res = B.prove_batch(tree, subset, params, pk, levels, flavour="evm", in_flight=in_flight)
        except Exception as ex:                      # carried to the other ranks as a count, below
            failure, res = 1, B.BatchResult()
            res.errors[-1] = repr(ex)
        torch.cuda.synchronize()
        cpu = cpu_seconds() - c1
        if world > 1:
            _beat("barrier after the batch")
            dist.barrier()
        dt = _all_max(time.perf_counter() - t1, world, coll_dev)
        done, errs, cpu_all = _all_sum([len(res.proofs), len(res.errors) + failure, cpu], world, coll_dev)
        made[0] += int(done)
        return int(done), int(errs), dt, res, cpu_all

    # warm-up and pre-sweep (short batches): every lane's plans, every worker thread's session; the best in-flight setting
    # of the pre-sweep runs the headline batch
    sweep = {}
    per = max(6 * world, min(96 * world, total // 4))
    candidates = sorted({max(1, args.batch_in_flight // 2), args.batch_in_flight}) if args.batch_in_flight > 0 else (1, 2, 4, 8, 16, 24, 32, 48, 64, 96)
    for in_flight in candidates:
        timed_batch(users[:3 * in_flight * world], in_flight)
        done, errs, dt, _, _ = timed_batch(users[:min(total, max(per, 6 * in_flight * world))], in_flight)
        sweep[str(in_flight)] = {"proofs": done, "errors": errs, "seconds": dt, "proofs_per_s": done / dt if dt else 0.0}
    best_in_flight = int(max(sweep, key=lambda f: sweep[f]["proofs_per_s"]))
    out["pre_sweep_by_in_flight"] = sweep
    out["in_flight"] = best_in_flight
    reps, last = [], None
    for _ in range(max(1, args.batch_repeats)):
        done, errs, dt, res, cpu = timed_batch(users, best_in_flight, headline=True)
        reps.append({"proofs": done, "errors": errs, "seconds": dt, "proofs_per_s": done / dt if dt else 0.0,
                     "host_cpu_ms_per_proof": cpu / max(1, done) * 1e3, "host_cores_busy": cpu / dt / world if dt else 0.0})
        last = res
    rates = sorted(r["proofs_per_s"] for r in reps)
    out["repeats"] = reps
    out["errors"] = sum(r["errors"] for r in reps)
    out["proofs_per_s_min_median_max"] = [rates[0], rates[len(rates) // 2], rates[-1]]
    out["proofs_per_s"] = rates[len(rates) // 2]
    out["seconds"] = sorted(r["seconds"] for r in reps)[len(reps) // 2]
    out["host_cpu_ms_per_proof"] = sorted(r["host_cpu_ms_per_proof"] for r in reps)[len(reps) // 2]
    out["host_cores_busy_per_gpu"] = sorted(r["host_cores_busy"] for r in reps)[len(reps) // 2]
    out["host"] = host_cpu_info()
    out["wait_sleep_us"] = last.wait_sleep_us
    out["proofs_made_in_run"] = made[0]
    assert out["errors"] == 0 and all(r["proofs"] == total for r in reps), out
    if rank == 0:   # checker leg, outside every timed region: the oracle's verifier on a sample of the proofs made
        from oracle import summa_verifier as SV
        ovk = oracle_vk(params, vk)
        sample = sorted(last.proofs)[:3]
        out["verified_sample"] = all(SV.verify(last.proofs[u][0], last.proofs[u][1], ovk) for u in sample) and bool(sample)
        out["note"] = ("gen_proof_solidity_calldata per user (Keccak transcript; EVERY proof is re-verified by the product's verifier "
                       "before it is handed back, as the reference's create_proof_checked does -- a rejected proof would count as an "
                       "error), witness synthesis included; proofs_total users dealt round-robin over the ranks; whole-job rate = "
                       "proofs of all ranks / max-over-ranks time; proofs_per_s = median of the repeats")
    return (out if rank == 0 else None), (tree, params, pk, vk)


def strong_scaling_extra(args, rank, world, coll_dev):
    """ONE MSM of 2^23 points, point-sharded over the N ranks (circuits_halo2_amd.distributed.sharded_msm): total work
    fixed as N grows, so the 1 -> 8 curve shows what the exchange step (one all_gather of 64-byte partials + a host sum)
    costs.  The input is 8 sub-shards of 2^20 (seeded by sub-shard), rank r holds the sub-shards [8 r / N, 8 (r + 1) / N):
    the same 2^23 points at every N, hence the same result point (reported, for comparison across the runs).  Each
    rank's partial is checked against <k, s> G of its own shard (bases are s_i G with known s_i) before the timing."""
    import torch
    import torch.distributed as dist
    import circuits_halo2_amd as sg
    from circuits_halo2_amd import arithmetic as A
    from circuits_halo2_amd.distributed import sharded_msm
    from circuits_halo2_amd.utils import random_fr_canonical
    log_total, subs = 23, 8
    if world > subs or subs % world:
        return {"skipped": f"world size {world} does not divide {subs} sub-shards"} if rank == 0 else None
    mine = range(subs * rank // world, subs * (rank + 1) // world)
    sub_n = 1 << (log_total - 3)
    sc = torch.cat([A.fr_to_montgomery(torch.from_numpy(random_fr_canonical(0xC0FFEE + 2 * j, sub_n)).cuda()) for j in mine])
    bs = torch.cat([A.fr_to_montgomery(torch.from_numpy(random_fr_canonical(0xC0FFEE + 2 * j + 1, sub_n)).cuda()) for j in mine])
    bases = A.g1_fixed_base_mul(bs)
    torch.cuda.synchronize()
    part = sg.best_multiexp(sc, bases)
    one = np.frombuffer((0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb).to_bytes(32, "little"), dtype=np.uint8)
    inner = A.eval_polynomial(A.fr_mul(sc, bs), one)                       # <k, s> = sum_i k_i s_i  (evaluation at 1)
    want = A.g1_fixed_base_mul(inner)
    ok_local = bool((part == want).all())
    del bs
    _beat("strong-scaling MSM", 120.0)
    sharded_msm(sc, bases)
    steps = max(3, min(args.steps, 10))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        point = sharded_msm(sc, bases)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = _all_max(time.perf_counter() - t0, world, coll_dev)
    ok = _all_sum([1.0 if ok_local else 0.0], world, coll_dev)[0] == world
    if rank != 0:
        return None
    return {"log_n_total": log_total, "points_per_gpu": sc.numel() // 32, "n_gpus": world, "steps": steps, "scaling": "strong",
            "ms_per_msm": dt / steps * 1e3, "points_per_s": (1 << log_total) * steps / dt,
            "result_x": bytes(point[:32])[::-1].hex(), "every_rank_partial_equals_inner_product_times_G": ok,
            "note": "one 2^23-point MSM sharded over the ranks (contiguous shards), ONE all_gather_into_tensor of 64-byte partials per MSM, "
                    "host sum of N points; result_x (Montgomery words, big-endian hex) is the same at every N"}


def cpu_oplist_baseline(k, cores):
    """restated-reference CPU op list of ONE k = 17 proof on this box's cores (context, never the target): the oracle's
    halo2-shaped best_multiexp / best_fft / evaluate_h blocks in the counts of SURVEY.md section 3.1 -- 16 MSM(2^k),
    9 iNTT(2^k), 9 coset NTT(2^(k+3)), 1 iNTT(2^(k+3)), the gate / permutation / lookup folds over 2^(k+3) rows
    [REF zk_prover/src/circuits/utils.rs:88,105: the reference times create_proof as a whole]"""
    from circuits_halo2_amd import mst_inclusion as M
    from oracle import oracle as O
    n, ext_k = 1 << k, k + 3
    ne = 1 << ext_k
    t = {}
    sc, bases = O.random_fr(901, n), O.fixed_base_mul(O.random_fr(902, n), cores)
    t0 = time.perf_counter()
    for _ in range(16):
        O.best_multiexp(sc, bases, cores)
    t["msm_16x2^k_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    for _ in range(9):
        co = O.lagrange_to_coeff(sc, k, cores)
    t["intt_9x2^k_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    for _ in range(9):
        ext = O.coeff_to_extended(co, k, ext_k, cores)
    t["coset_ntt_9x2^(k+3)_ms"] = (time.perf_counter() - t0) * 1e3
    cols = [ext, O.random_fr(903, ne), O.random_fr(904, ne)]
    pick = lambda i: cols[i % 3]
    beta, gamma, theta, y = (O.random_fr(910 + i, 1) for i in range(4))
    O.set_quotient_threads(cores)
    try:
        t0 = time.perf_counter()
        v = O.quotient_gates(np.zeros(32 * ne, dtype=np.uint8), M.gate_graph(2).as_dict(), [pick(i) for i in range(11)],
                             [pick(i + 1) for i in range(3)], [pick(2)], M.gate_challenges(5), beta, gamma, theta, y, k, ext_k)
        v = O.quotient_permutation(v, [pick(0), pick(1)], [pick(i) for i in range(6)], [pick(i + 1) for i in range(6)], 4, pick(0), pick(1),
                                   pick(2), beta, gamma, y, k, ext_k, 6)
        v = O.quotient_lookup(v, pick(0), pick(1), pick(2), pick(0), pick(1), pick(2), pick(0), pick(1), beta, gamma, y, k, ext_k)
        t["evaluate_h_2^(k+3)_rows_ms"] = (time.perf_counter() - t0) * 1e3
    finally:
        O.set_quotient_threads(1)
    t0 = time.perf_counter()
    O.extended_to_coeff(O.divide_by_vanishing_poly(v, k, ext_k), k, ext_k, cores)
    t["intt_1x2^(k+3)_ms"] = (time.perf_counter() - t0) * 1e3
    return {"total_ms": sum(t.values()), "parts_ms": {a: round(b, 1) for a, b in t.items()}, "cores": cores,
            "affinity_cores": len(os.sched_getaffinity(0)), "nproc": os.cpu_count(), "kind": "port", "label": "restated-reference CPU op list (MSM + NTT + evaluate_h of one proof; grand products, "
                                     "evaluations and the multi-open's polynomial arithmetic not included)"}


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=9)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip NTT / op-list extras")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1: nccl (= RCCL, one GPU per rank); gloo rehearses the multi-rank logic on a "
                         "box with fewer GPUs than ranks (host-side collectives, ranks share GPUs round-robin)")
    ap.add_argument("--in-flight", type=int, default=3, help="steps in flight per GPU: host threads issuing MSMs (library lanes); 1 = strictly one after the other")
    ap.add_argument("--batch-proofs", type=int, default=1024, help="k = 17 inclusion proofs IN TOTAL in the batch extra, dealt over the ranks "
                                                                  "(BASELINE configs[4]: 1024 users; 0 = skip)")
    ap.add_argument("--strong-only", action="store_true", help="of the extras, run only the strong-scaling MSM (tests)")
    ap.add_argument("--batch-repeats", type=int, default=3, help="timed repeats of the whole batch (min / median / max reported)")
    ap.add_argument("--wall-limit", type=float, default=540.0,
                    help="seconds after which the run gives up and prints a JSON line with `error` and what it had (below the driver's own limit)")
    ap.add_argument("--cpu-share", type=int, default=0,
                    help="run on the first N of the CPUs this process may use (os.sched_setaffinity, set before anything touches the GPU, so "
                         "every thread of the HIP runtime and of the library inherits it): what one rank of N_total / N ranks sharing the host gets")
    ap.add_argument("--batch-only", action="store_true", help="only the proof batch (the child runs of the --cpu-share rehearsal): prints the batch extra's dict")
    ap.add_argument("--batch-in-flight", type=int, default=0, help="proofs in flight for the batch (0 = pre-sweep picks)")
    ap.add_argument("--no-cpu-share-sweep", action="store_true", help="skip the CPU-share rehearsal of the batch (three child runs)")
    ap.add_argument("--acc-log", default="", help="profiling: write the library's msm_accumulate launch log (sg_msm_launch_log) to this file at the end "
                                                  "of the headline steps / of the batch, so that a kernel trace of this run attributes every launch to its job exactly")
    args = ap.parse_args()

    if args.cpu_share > 0:
        allowed = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, allowed[:max(1, min(args.cpu_share, len(allowed)))])

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare (`python bench.py --gpus N`): start the N ranks ourselves, one process per GPU, BEFORE anything
        # in this process touches the GPU (a child process, never an exec of a process that has initialised HIP);
        # rank 0 of the children prints the JSON line, which passes through -- or the parent says what went wrong
        sys.exit(run_ranks_and_relay(args, sys.argv[1:]))

    # HIP's default of 4 hardware queues is kept: more of them did not help the MSM lanes and made the proof batch slower and
    # erratic (profiles/r02_sweeps/hw_queues.txt)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if args.backend == "gloo" and torch.cuda.device_count() >= 1:
        local_rank %= torch.cuda.device_count()          # rehearsal: more ranks than GPUs
    if torch.cuda.device_count() < max(1, min(world, local_rank + 1)):
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"   # where the small tensors of the timing collectives live
    global _FS
    fs = _FS = FailSafe(rank, args, args.wall_limit)
    fs.arm()
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _beat("init_process_group", 100.0)
        # a rank that never arrives must not cost the others the driver's whole time limit (the default is ten minutes)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=90))
        else:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=90))
    _beat("set-up", 300.0)

    import circuits_halo2_amd as sg
    from circuits_halo2_amd import ffi
    from circuits_halo2_amd.arithmetic import fr_to_montgomery, g1_fixed_base_mul
    from circuits_halo2_amd.distributed import exchange_partials_many
    from circuits_halo2_amd.utils import DEFAULT_SEED, random_fr_canonical
    ffi.check(sg.lib().sg_init(local_rank))
    for kv in filter(None, os.environ.get("SG_PARAMS", "").split(",")):  # e.g. SG_PARAMS=msm.log_seg=6
        name, val = kv.split("=")
        ffi.check(sg.lib().sg_set_param(name.encode(), int(val)))

    def dump_acc_log(region):
        if args.acc_log and rank == 0:
            with open(args.acc_log, "w") as f:
                json.dump({"region": region, "launches": ffi.msm_launch_log()}, f)

    if args.acc_log:
        ffi.set_param("msm.acc_log", 1)
    if args.batch_only:      # a child run of the CPU-share rehearsal: the batch extra alone, its dict as the line
        got, _ = batch_extra(args, rank, world, coll_dev)
        dump_acc_log("process start .. end of the batch")
        if rank == 0:
            got["cpu_share"] = args.cpu_share
            fs.emit(got)
        sys.stdout.flush()
        return got           # (a normal exit: a profiler attached to this process writes its database at exit)

    n = 1 << args.log_n
    # synthetic inputs, generated per rank from rank-dependent seeds, resident in HBM
    scal = fr_to_montgomery(torch.from_numpy(random_fr_canonical(DEFAULT_SEED + rank, n)).cuda())
    base_scal = fr_to_montgomery(torch.from_numpy(random_fr_canonical(0x7A55 + rank, n)).cuda())
    bases = g1_fixed_base_mul(base_scal)  # s_i * G: valid, distinct curve points
    torch.cuda.synchronize()

    from concurrent.futures import ThreadPoolExecutor
    import threading
    tls = threading.local()

    pipelined_timings = None      # a list while the steps of a region record their own HIP-event timings (after the timed region)

    def partial():
        """this rank's shard of one step: a whole 2^log_n MSM (digits .. host tail), result = the 64-byte point"""
        if not hasattr(tls, "stream"):
            tls.stream = torch.cuda.Stream()
        with torch.cuda.stream(tls.stream):
            if pipelined_timings is not None:
                point, tm = sg.best_multiexp(scal, bases, timings=True)
                pipelined_timings.append(tm)
                return point
            return sg.best_multiexp(scal, bases)

    def run_steps(count, in_flight):
        """`count` steps; with in_flight > 1 the MSMs of consecutive steps are issued from that many host threads (each call
        takes its own lane of the library: streams, engines, work space); the exchange step of the `count` MSMs is ONE
        collective on this thread after the last partial has arrived (distributed.exchange_partials_many: the partials
        are 64 bytes each and already on the host; RCCL's kernels cannot raise their wave priority and would crawl beside
        the chained accumulations, so no collective runs while the device is saturated)"""
        if in_flight <= 1:
            return exchange_partials_many([partial() for _ in range(count)])
        # the threads take step numbers from one counter (three submissions, not `count`: the main thread holds the interpreter
        # lock while it submits, and the pool's threads start only when it lets go)
        import itertools
        ticket, results, failure = itertools.count(), [None] * count, []
        ready = [threading.Event() for _ in range(count)]

        def issue():
            while not failure:
                i = next(ticket)
                if i >= count:
                    return
                try:
                    results[i] = partial()
                except BaseException as ex:      # noqa: BLE001 -- handed to the main thread, which raises it
                    failure.append(ex)
                    for ev in ready:
                        ev.set()
                    return
                ready[i].set()
        futures = [pool.submit(issue) for _ in range(in_flight)]
        for i in range(count):
            ready[i].wait()
            if failure:
                break
        for f in futures:
            f.result()
        if failure:
            raise failure[0]
        return exchange_partials_many(results)

    def timed(count, in_flight):
        _beat("timed steps", 75.0 + 0.05 * count)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        results = run_steps(count, in_flight)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        return _all_max(dt, world, coll_dev), results

    in_flight = max(1, args.in_flight)
    if world > 1 and "host.wait_sleep_us" not in os.environ.get("SG_PARAMS", ""):
        # several ranks share the host's CPU quota: their waiting threads sleep between polls instead of yielding (neutral at
        # N = 1: profiles/r04_sweeps/headline_wait_modes.txt)
        ffi.check(sg.lib().sg_set_param(b"host.wait_sleep_us", 25))
    pool = ThreadPoolExecutor(max_workers=in_flight, initializer=ffi.bind_thread)   # a new thread's current device is 0
    run_steps(2 * in_flight, in_flight)      # set-up, not a step: every lane allocates its work space (first call per lane)
    run_steps(args.warmup, in_flight)
    dt, results = timed(args.steps, in_flight)
    result = results[-1]
    assert all((r == result).all() for r in results), "steps of one input must agree"
    fs.partial.update({"value": world * n * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "n_gpus": world})
    # the SAME regime once more, untimed, with every step's own HIP events (sg_msm_timings: the chained accumulation launch from
    # behind its wait for the previous one): the per-kernel figure of the regime `value` is quoted on, for `roofline`
    # (every rank runs it -- the region ends in the ranks' collective -- rank 0 reports its own launches)
    if in_flight > 1:
        pipelined_timings = []
        run_steps(max(args.steps, 2 * in_flight), in_flight)
        pipelined_reps, pipelined_timings = pipelined_timings[in_flight:], None    # (the first ones start on an idle device)
    else:
        pipelined_reps = None
    # the same steps strictly one after the other (the latency of one MSM, round 1's headline)
    run_steps(2, 1)
    # (an extra, not the headline: the better of two repeats of `steps` blocking calls -- one hiccup of the host in a 35 ms region
    # reads as +25 % otherwise; seen once in thirty runs)
    dt_seq = min(timed(args.steps, 1)[0] for _ in range(2))
    fs.partial["sequential_ms_per_step"] = dt_seq / args.steps * 1e3
    # N > 1: what ONE blocking point-sharded MSM costs end to end -- the shard's MSM, ONE collective of its own, the host sum of N
    # points (distributed.sharded_msm) -- beside the amortised exchange of the timed region: the latency point of a scaling run
    percall = None
    if world > 1:
        from circuits_halo2_amd.distributed import sharded_msm
        calls = max(3, min(args.steps, 10))
        sharded_msm(scal, bases)
        _beat("per-call sharded MSM", 75.0)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            p_call = sharded_msm(scal, bases)
        torch.cuda.synchronize()
        dist.barrier()
        dt_call = _all_max(time.perf_counter() - t0, world, coll_dev)
        assert (p_call == result).all(), "the per-call exchange must give the point of the amortised one"
        percall = {"ms_per_msm": dt_call / calls * 1e3, "calls": calls, "points_per_gpu": n, "value": world * n * calls / dt_call,
                   "note": "one collective (all_gather of N x 64 B) per MSM, nothing in flight beside it: the latency an unmodified caller of a "
                           "point-sharded best_multiexp sees at this N; `value` amortises the exchange over the timed region instead"}
        fs.partial["per_call_sharded_msm_ms"] = percall["ms_per_msm"]
    pool.shutdown()
    _beat("extras", args.wall_limit)
    phase_reps = None
    if rank == 0:   # per-phase HIP-event timings of the same MSM, taken here (the extras below fill HBM and caches with other data)
        sg.best_multiexp(scal, bases, timings=True)
        phase_reps = [sg.best_multiexp(scal, bases, timings=True)[1] for _ in range(5)]
    dump_acc_log("process start .. the five lone timed MSMs after the headline steps")

    # ---- the line's headline part, complete BEFORE any extra runs: if an extra hangs or the wall-clock limit comes, the
    # watchdog still has a whole, valid line to print (FailSafe.base_line)
    line = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * n * args.steps / dt
        line = {
            "metric": "msm_points_per_sec", "value": value, "unit": "points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "dtype_note": "254-bit Montgomery integers: 8 x u32 words in memory, 9 x 29-bit limbs with 64-bit accumulators in registers",
            "data": "synthetic",
            "config": {"workload": f"standalone BN254 G1 MSM, 2^{args.log_n} uniform Fr scalars x synthetic-SRS "
                                   f"affine points per GPU (BASELINE configs[1])",
                       "points_per_gpu": n, "sharding": "point-sharded; exchange = ONE all_gather of steps x 64-B partials per timed region + host sums" if world > 1 else "none",
                       "backend": ("nccl (RCCL)" if args.backend == "nccl" else "gloo (rehearsal: ranks share GPUs)") if world > 1 else None,
                       "steps_in_flight": in_flight,
                       "step": "one whole MSM per GPU (digits, sort, accumulate, reduce, host tail; result = the 64-byte point); "
                               "consecutive steps are issued from steps_in_flight host threads, each call on its own lane of the library"},
            "sequential": {"ms_per_step": dt_seq / args.steps * 1e3, "value": world * n * args.steps / dt_seq,
                           "note": "the same steps strictly one after the other (--in-flight 1): the latency of one MSM"},
        }
        if world > 1:
            # what the collectives library itself saw (the driver's "did RCCL see N ranks" check): the process group's size and backend
            line["config"]["ranks_seen_by_process_group"] = dist.get_world_size()
            line["config"]["rccl_ranks_seen"] = dist.get_world_size() if args.backend == "nccl" else None
            line["config"]["per_call_sharded_msm_ms"] = percall["ms_per_msm"]
            line["sharded_msm_per_call"] = percall
        # ---- roofline of the dominant kernel (msm_accumulate), HIP events on its stream
        reps = phase_reps
        acc_ms = float(np.mean([r["accumulate_ms"] for r in reps]))
        alg_bytes = MSM_BYTES_PER_PAIR * n
        achieved = alg_bytes / (acc_ms * 1e-3) / 1e9
        lone = {"regime": "one MSM at a time (the device to itself: three waves per SIMD)", "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                "launch_ms": acc_ms, "accumulate_threads": reps[0]["accumulate_threads"], "launches_timed": len(reps)}
        lone.update(pmc_traffic("sg::msm_accumulate", args.log_n, reps[0]["accumulate_threads"], n))
        if pipelined_reps:
            # the regime the headline runs in: steps_in_flight MSMs in flight, accumulations chained one behind the other (two waves
            # per SIMD, the other MSMs' sort / reduction kernels running under them at wave priority 3)
            p_ms = float(np.mean([r["accumulate_ms"] for r in pipelined_reps]))
            p_achieved = alg_bytes / (p_ms * 1e-3) / 1e9
            line["roofline"] = {"kernel": "msm_accumulate", "bound": "hbm", "achieved": p_achieved, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": p_achieved / HBM_PEAK_GBS, "traffic": None,
                                "launch_ms": p_ms, "algorithmic_bytes": alg_bytes,
                                "regime": f"the headline's own: {in_flight} MSMs in flight, chained accumulation launches (two waves per SIMD) timed by "
                                          "their own HIP events from behind the chain's wait, in an untimed repeat of the timed region",
                                "accumulate_threads": pipelined_reps[0]["accumulate_threads"], "launches_timed": len(pipelined_reps),
                                "launch_ms_min_max": [float(min(r["accumulate_ms"] for r in pipelined_reps)), float(max(r["accumulate_ms"] for r in pipelined_reps))],
                                "share_of_step": p_ms / ms_per_step,
                                "lone": lone}
            line["roofline"].update(pmc_traffic("sg::msm_accumulate", args.log_n, pipelined_reps[0]["accumulate_threads"], n))
        else:
            line["roofline"] = {"kernel": "msm_accumulate", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                                "launch_ms": acc_ms, "algorithmic_bytes": alg_bytes, "regime": lone["regime"]}
            line["roofline"].update({k_: v_ for k_, v_ in lone.items() if k_.startswith("traffic")})
        line["msm_phases_ms"] = {k: float(np.mean([r[k] for r in reps])) for k in
                                 ("digits_ms", "sort_ms", "order_ms", "accumulate_ms", "reduce_ms", "total_ms")}
        line["msm_phases_ms"].update({k: reps[0][k] for k in ("window_bits", "windows", "tasks", "max_bucket", "accumulate_threads")})
        # integer-ALU view (MSM is VALU-bound): mixed adds * 10 products * ~560 VALU instr
        adds = reps[0]["windows"] * n
        mads = adds * MADS_PER_MIXED_ADD
        line["alu"] = {"bucket_adds_per_launch": adds, "mixed_adds_per_s": adds / (acc_ms * 1e-3),
                       "mad_u64_u32_lane_ops_per_s": mads / (acc_ms * 1e-3), "mad_peak_lane_ops_per_s": MAD_PEAK,
                       "frac_of_mad_peak": mads / (acc_ms * 1e-3) / MAD_PEAK,
                       "note": "integer-VALU view: a mixed add is 6 products + 2 squarings + 1 fused two-product "
                               "reduction on 9x29-bit limbs = 1467 v_mad_u64_u32 of 2 170 VALU instructions per addition "
                               "(SQ_INSTS_VALU of the launch / additions, profiles/r03z_insts.json: the same for the persistent launch of round 3 as for the grid of round 2; 2 383 before the products "
                               "became one multiply-add chain per column and the sums between them shared their carry steps); peak = "
                               "measured v_mad_u64_u32 issue rate (profiles/r01_microbench_instr_rates.txt)"}

        # hardware-counter view of the same claim, from the newest committed VALU pass (profiles/*_valu.json)
        import glob as _glob
        for path in sorted(_glob.glob(os.path.join(ROOT, "profiles", "*_valu.json")), reverse=True):
            pg = json.load(open(path)).get("per_grid", {})
            exact = pg.get(f"sg::msm_accumulate@grid{reps[0]['accumulate_threads']}@job{n}")
            cands = [(0, exact)] if exact else [(abs(int(k.split("@grid")[1]) - reps[0]["accumulate_threads"]), v) for k, v in pg.items()
                                                if k.startswith("sg::msm_accumulate@") and "@job" not in k]
            if cands:
                v = min(cands, key=lambda kv: kv[0])[1]
                line["alu"].update({"valu_busy_pct": v.get("VALUBusy_avg"), "valu_lane_utilization_pct": v.get("VALUUtilization_avg"),
                                    "valu_source": os.path.basename(path)})
                break

        fs.base_line = line

    # the extras that every rank takes part in.  An extra never costs the headline line, and a rank that fails locally
    # still meets the others: after each extra the ranks agree on a failure flag (all_reduce), never on a bare barrier
    # that a failed rank would not reach
    def collective_extra(fn):
        res, failed = None, 0.0
        try:
            res = fn()
        except Exception as ex:
            res, failed = {"error": repr(ex)}, 1.0
        any_failed = _all_max(failed, world, coll_dev) > 0
        return res, any_failed

    strong_line = None
    if not args.no_extras and args.log_n >= 20:
        strong_line, _ = collective_extra(lambda: strong_scaling_extra(args, rank, world, coll_dev))
        if rank != 0:
            strong_line = None
        torch.cuda.empty_cache()
    batch_line, k17 = None, None
    if args.batch_proofs > 0 and not args.no_extras and args.log_n >= 20:
        got, any_failed = collective_extra(lambda: batch_extra(args, rank, world, coll_dev))
        if isinstance(got, tuple):
            batch_line, k17 = got
        else:
            batch_line = got if rank == 0 else None
        if any_failed and rank == 0 and (batch_line is None or "error" not in batch_line):
            batch_line = {"error": "another rank failed in the batch extra"}
        if any_failed:
            k17 = None
        if rank == 0 and isinstance(batch_line, dict):
            fs.partial["batch_k17"] = {k_: batch_line.get(k_) for k_ in ("proofs_per_s", "proofs_total", "errors", "error", "in_flight") if k_ in batch_line}
    if world > 1:
        # the last collective of the run: what follows is rank 0's own (the line, the local extras); the other ranks leave
        # without waiting for it, so no rank sits in a barrier while rank 0 times a CPU baseline
        _beat("final barrier")
        dist.barrier()
        dist.destroy_process_group()
    _beat("rank 0: local extras and the line", args.wall_limit)

    if rank == 0:
        # throughput mode: the same MSM issued as a batch of 8 (fused / pipelined jobs)
        sg.best_multiexp_batch([(scal, bases)] * 8)   # warms both engines' work spaces
        torch.cuda.synchronize()
        bdt = 1e9
        for _ in range(3):
            t1 = time.perf_counter()
            outs = sg.best_multiexp_batch([(scal, bases)] * 8)
            torch.cuda.synchronize()
            bdt = min(bdt, time.perf_counter() - t1)
        assert all((o == result).all() for o in outs) or world > 1
        line["batched"] = {"msms": 8, "ms_per_msm": bdt / 8 * 1e3, "points_per_s": 8 * n / bdt}

        if not args.no_extras and not args.strong_only and world == 1 and args.log_n >= 20:
            try:  # extras never cost the headline line
                line["ntt"] = {}
                for lg in (17, 22):
                    a = fr_to_montgomery(torch.from_numpy(random_fr_canonical(DEFAULT_SEED + 100 + lg, 1 << lg)).cuda())
                    ms = C.c_float(0)
                    ffi.check(sg.lib().sg_time_ntt_dev(ffi.dev_ptr(a), C.c_uint32(lg), 20, C.byref(ms)))
                    gbs = NTT_BYTES_PER_ELEM * (1 << lg) / (ms.value * 1e-3) / 1e9
                    line["ntt"][f"2^{lg}"] = {"ms": ms.value, "elements_per_s": (1 << lg) / (ms.value * 1e-3),
                                              "algorithmic_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}
                    del a
                # k = 17 proof op list: 16 MSM(2^17) + 9 iNTT(2^17) + 9 NTT(2^20) + 1 iNTT(2^20)
                k = 17
                s17, b17 = scal[: 32 << k], bases[: 64 << k]
                sg.best_multiexp_batch([(s17, b17)] * 16)
                torch.cuda.synchronize()
                msm17 = 1e9
                for _ in range(3):   # best of 3: a single sample right after the NTT loops is noisy
                    t1 = time.perf_counter()
                    sg.best_multiexp_batch([(s17, b17)] * 16)
                    torch.cuda.synchronize()
                    msm17 = min(msm17, (time.perf_counter() - t1) * 1e3)
                a20 = scal[: 32 << 20].clone()
                ms20 = C.c_float(0)
                ffi.check(sg.lib().sg_time_ntt_dev(ffi.dev_ptr(a20), C.c_uint32(20), 10, C.byref(ms20)))
                ntt_ms = 9 * line["ntt"]["2^17"]["ms"] + 10 * ms20.value
                # the same 16 commitments the way a prover with a resident SRS issues them: ParamsKZG.commit_batch
                # over the precomputed window table (sg_srs_precompute; one-off cost reported beside it)
                b17 = b17.cpu().numpy()
                params = sg.ParamsKZG(k, b17, b17)
                generic = params.commit_batch([s17] * 16)
                t1 = time.perf_counter()
                params.precompute(0)
                pre_ms = (time.perf_counter() - t1) * 1e3
                fixed = params.commit_batch([s17] * 16)
                assert (fixed == generic).all(), "fixed-base commit differs from the generic MSM"
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    params.commit_batch([s17] * 16)
                torch.cuda.synchronize()
                msm17_fixed = (time.perf_counter() - t1) / 5 * 1e3
                t1 = time.perf_counter()
                for _ in range(10):
                    params.commit(s17)
                one17_fixed = (time.perf_counter() - t1) / 10 * 1e3
                params.free()
                # the headline's MSM the way a prover holds its SRS: resident in HBM with the fixed-base window table
                # (sg_srs_upload + sg_srs_precompute -> ParamsKZG.commit); same scalars, same bases, same result
                try:
                    host_bases = bases.cpu().numpy()
                    p20 = sg.ParamsKZG(args.log_n, host_bases, host_bases)
                    t1 = time.perf_counter()
                    p20.precompute(0)
                    pre20_ms = (time.perf_counter() - t1) * 1e3
                    same = bool((p20.commit(scal) == sg.best_multiexp(scal, bases)).all())
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(10):
                        p20.commit(scal)
                    seq20 = (time.perf_counter() - t1) / 10 * 1e3

                    def one_commit(_):
                        if not hasattr(tls, "stream"):
                            tls.stream = torch.cuda.Stream()
                        with torch.cuda.stream(tls.stream):
                            return p20.commit(scal)
                    with ThreadPoolExecutor(max_workers=3, initializer=ffi.bind_thread) as tp:
                        list(tp.map(one_commit, range(6)))
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        list(tp.map(one_commit, range(30)))
                        torch.cuda.synchronize()
                        pipe20 = (time.perf_counter() - t1) / 30 * 1e3
                    p20.free()
                    line["msm_resident_srs"] = {"log_n": args.log_n, "ms_per_msm_3_in_flight": pipe20, "ms_per_msm_sequential": seq20,
                                                "points_per_s": n / (pipe20 * 1e-3), "same_result_as_generic": same,
                                                "window_table_once_ms": pre20_ms,
                                                "note": "ParamsKZG.commit against the SRS resident in HBM with its fixed-base window table "
                                                        "(W x n affine points, built once per SRS): all digits of a scalar land in one bucket "
                                                        "set.  Context only: the headline above is the generic MSM over arbitrary bases"}
                except Exception as ex:
                    line["msm_resident_srs"] = {"error": repr(ex)}
                line["proof_oplist_k17"] = {"msm16x2^17_ms": msm17, "msm16x2^17_fixed_base_ms": msm17_fixed,
                                            "single_commit_2^17_fixed_base_ms": one17_fixed,
                                            "srs_precompute_once_ms": pre_ms, "ntt_19_ms": ntt_ms,
                                            "sum_ms": msm17_fixed + ntt_ms, "sum_generic_msm_ms": msm17 + ntt_ms,
                                            "rows_per_s": (1 << k) / ((msm17_fixed + ntt_ms) * 1e-3),
                                            "note": "MSM+NTT kernel sum only; sum_ms uses the resident-SRS commit path "
                                                    "(fixed-base window table), sum_generic_msm_ms arbitrary bases"}

                # witness side (row W): Merkle sum tree of 2^20 users, 1 currency (the reference bench's LEVELS = 20
                # shape, zk_prover/benches/full_solvency_flow.rs:13-16): Poseidon leaves + 20 levels on the device
                depth, nc = 20, 1
                d_users, d_bals = scal[: 32 << depth], scal[: (32 * nc) << depth].clone()
                d_bals.view(-1, 32)[:, 8:] = 0                      # 64-bit balances (N_BYTES = 8), Montgomery form not needed for timing
                nodes = (2 << depth) - 1
                d_h = torch.empty(32 * nodes, dtype=torch.uint8, device="cuda")
                d_b = torch.empty(32 * nodes * nc, dtype=torch.uint8, device="cuda")
                build = lambda: ffi.check(sg.lib().sg_mst_build_dev(ffi.dev_ptr(d_users), ffi.dev_ptr(d_bals), C.c_uint32(depth),
                                                                    C.c_uint32(nc), ffi.dev_ptr(d_h), ffi.dev_ptr(d_b), None))
                build(); torch.cuda.synchronize()
                t1 = time.perf_counter()
                build(); torch.cuda.synchronize()
                mst_ms = (time.perf_counter() - t1) * 1e3
                line["witness_mst_2^20"] = {"ms": mst_ms, "poseidon_permutations_per_s": ((1 << depth) * (1 + nc) + ((1 << depth) - 1) * (2 + nc)) / (mst_ms * 1e-3),
                                            "note": "leaf = H(username, balances) = 1 + nc absorptions, middle = H(sum balances, left, right) = 2 + nc; "
                                                    "one permutation per absorbed element at rate 1"}
                del d_h, d_b, d_bals
                # the whole create_proof op schedule at k = 17 with its Fiat-Shamir sync points (tools/proof_flow.py):
                # 16 commitments in 6 groups, 9 + 9 + 1 transforms, evaluate_h (gates + permutation + lookup), 35
                # evaluations, multi-open -- synthetic witness and a stand-in gate program, every op through the C ABI
                try:
                    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
                    from proof_flow import run_flow
                    del a20
                    torch.cuda.empty_cache()
                    flow = run_flow(17, n_gates=12, reps=3, overlap=False)
                    flow_ov = run_flow(17, n_gates=12, reps=3, overlap=True)
                    cpp = {}
                    exe = os.path.join(ROOT, "tools", "proof_flow_cpp")
                    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k_.startswith(("ROCPROF", "ROCP_")) for k_ in os.environ)
                    if os.path.exists(exe) and not profiled:   # the same schedule driven from C++ over the C ABI (own process / context)
                        import subprocess
                        torch.cuda.empty_cache()
                        r = subprocess.run([exe, "17", "12", "5"], capture_output=True, text=True, timeout=300)
                        if r.returncode == 0:
                            cpp = json.loads(r.stdout.strip().splitlines()[-1])
                    line["proof_flow_k17"] = {"ms": flow["total"], "ms_cpp_driver": cpp.get("total"),
                                              "phases_ms_cpp_driver": {k_: v_ for k_, v_ in cpp.items() if k_[0].isdigit()}, "phases_ms": {k_: round(v_, 3) for k_, v_ in flow.items() if k_ != "total"},
                                              "ms_multi_stream": flow_ov["total"],
                                              "rows_per_s": (1 << 17) / (flow["total"] * 1e-3),
                                              "note": "synthetic create_proof-shaped schedule (MstInclusion column/argument counts, "
                                                      "12 stand-in Poseidon-round gates = 132 products per row; the reference's generated verifier evaluates 19 gate "
                                                      "expressions with 127 multiplications per point, InclusionVerifier.sol:495-902), host syncs at the 6 challenge points, "
                                                      "Python/ctypes driver overhead included; ms_multi_stream: transforms of a phase, the "
                                                      "three grand products and the rotation sets of the multi-open on side streams"}
                except Exception as ex:  # the flow is an extra: never lose the headline line over it
                    line["proof_flow_k17"] = {"error": repr(ex)}
                # a REAL proof at k = 17 through the reference's API (circuits_halo2_amd/api.py): MstInclusionCircuit<20,2,8> in the
                # reference's own floor plan, the inclusion witness of one user of the device-resident 2^20-user snapshot;
                # the timed proof itself is verified (product verifier AND the oracle's, outside the timed region)
                profiled17 = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k_.startswith(("ROCPROF", "ROCP_")) for k_ in os.environ)
                try:
                    from circuits_halo2_amd import api as _api, prover as _prover, verifier as _verifier
                    torch.cuda.empty_cache()
                    if k17 is None:
                        tree17 = snapshot_tree(20, 2)
                        params17, pk17, vk17 = _api.generate_setup_artifacts(17, None, _api.MstInclusionCircuit.init_empty(20, 2, 8))
                        params17.precompute()
                    else:
                        tree17, params17, pk17, vk17 = k17
                    t1 = time.perf_counter()
                    host_circuit = _api.MstInclusionCircuit.init(tree17.generate_proof(5), 20)
                    _api._advice_columns(pk17, host_circuit)
                    torch.cuda.synchronize()
                    witness_host_ms = (time.perf_counter() - t1) * 1e3     # Merkle proof to the host + synthesize with Python integers
                    _api._advice_columns(pk17, _api.MstInclusionCircuit.init_from_tree(tree17, 6))   # warm: program upload
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    circuit17 = _api.MstInclusionCircuit.init_from_tree(tree17, 5)
                    inst17 = circuit17.instances()[0]
                    adv17 = _api._advice_columns(pk17, circuit17)
                    torch.cuda.synchronize()
                    witness_ms = (time.perf_counter() - t1) * 1e3          # synthesize on the device (sg_mst_inclusion_witness_dev)
                    assert inst17 == host_circuit.instances()[0]
                    _prover.create_proof(params17, pk17, adv17, inst17)
                    ms17, proof17 = 1e9, None
                    for _ in range(3):
                        torch.cuda.synchronize(); t1 = time.perf_counter()
                        pr = _prover.create_proof(params17, pk17, adv17, inst17)
                        d_ = (time.perf_counter() - t1) * 1e3
                        if d_ < ms17:
                            ms17, proof17 = d_, pr
                    phases17 = {}
                    _prover.create_proof(params17, pk17, adv17, inst17, timings=phases17)
                    t1 = time.perf_counter()
                    ok_product = _verifier.verify_proof(params17, vk17, proof17, inst17, "evm")
                    verify_ms = (time.perf_counter() - t1) * 1e3
                    t1 = time.perf_counter()
                    calldata = _api.gen_proof_solidity_calldata(params17, pk17, _api.MstInclusionCircuit.init_from_tree(tree17, 6))
                    calldata_ms = (time.perf_counter() - t1) * 1e3
                    from oracle import summa_verifier as _SV     # checker leg
                    ok_oracle = _SV.verify(proof17, inst17, oracle_vk(params17, vk17)) and _SV.verify(calldata[0], calldata[1], oracle_vk(params17, vk17))
                    t1 = time.perf_counter()
                    blake = _api.full_prover(params17, pk17, circuit17, [inst17])
                    blake_ms = (time.perf_counter() - t1) * 1e3
                    ok_blake = _api.full_verifier(params17, vk17, blake, [inst17]) and _SV.verify(blake, inst17, oracle_vk(params17, vk17), flavour="blake2b")
                    cpp17 = {}
                    exe17 = os.path.join(ROOT, "tools", "create_proof_cpp")
                    if os.path.exists(exe17) and not profiled17:   # the same prover as compiled host code (include/summa_prover.hpp), own process
                        import subprocess, tempfile
                        with tempfile.TemporaryDirectory() as td:
                            _prover.export_bundle(os.path.join(td, "bundle.bin"), params17, pk17, adv17, inst17)
                            r = subprocess.run([exe17, os.path.join(td, "bundle.bin"), os.path.join(td, "proof.bin"), "8"],
                                               capture_output=True, text=True, timeout=300)
                            if r.returncode == 0:
                                cpp17 = json.loads(r.stdout.strip().splitlines()[-1])
                                cpp17["verified"] = bool(_SV.verify(open(os.path.join(td, "proof.bin"), "rb").read(), inst17, oracle_vk(params17, vk17)))
                    line["create_proof_k17"] = {"ms": ms17, "ms_cpp_driver": cpp17.get("create_proof_ms"), "verified": bool(ok_product and ok_oracle),
                                                "verified_cpp_driver_proof": cpp17.get("verified"),
                                                "gen_proof_solidity_calldata_ms": calldata_ms, "full_prover_blake2b_ms": blake_ms,
                                                "blake2b_proof_bytes": len(blake), "blake2b_verified": bool(ok_blake),
                                                "witness_synthesis_ms": witness_ms, "witness_synthesis_host_python_ms": witness_host_ms,
                                                "verify_ms_product": verify_ms,
                                                "phases_ms_cpp_driver_synchronised": {k_: v_ for k_, v_ in cpp17.items() if k_[0].isdigit()},
                                                "proof_bytes": len(proof17), "rows_per_s": (1 << 17) / ((cpp17.get("create_proof_ms") or ms17) * 1e-3),
                                                "phases_ms_synchronised": {k_: round(v_, 2) for k_, v_ in phases17.items()},
                                                "note": "whole create_proof (EVM transcript, SHPLONK) for MstInclusionCircuit<20,2,8> at k = 17 in the reference's own floor plan, "
                                                        "inclusion witness of one user of a device-resident 2^20-user snapshot, wall clock incl. the host glue; "
                                                        "ms: Python driver, best of 3, and THAT proof is the one verified (product verifier + oracle verifier); "
                                                        "ms_cpp_driver: include/summa_prover.hpp in its own process, best of 8; gen_proof_solidity_calldata_ms: "
                                                        "the backend's call = witness synthesis + proof + immediate re-verification"}
                    if not args.no_cpu:
                        from oracle import oracle as _O
                        # every host core this process may run on (sched_getaffinity), as SURVEY.md section 8(d) asks; the 16-thread
                        # figure of the earlier rounds beside it when the box offers more
                        line["create_proof_k17"]["cpu_baseline_oplist_ms"] = cpu_oplist_baseline(17, _O.ncpu())
                        if _O.ncpu() > 16:
                            line["create_proof_k17"]["cpu_baseline_oplist_ms_16_threads"] = cpu_oplist_baseline(17, 16)
                    if k17 is None:
                        params17.free()
                except Exception as ex:
                    line["create_proof_k17"] = {"error": repr(ex)}
                # the reference circuit itself (its own floor plan, verifying key and SRS; csv/entry_16.csv user 0, k = 11):
                # key generation reproduces the contract's 17 commitments, create_proof from the C++ driver
                try:
                    import subprocess, tempfile
                    from circuits_halo2_amd import mst_inclusion as _M, prover as _pv
                    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree, keccak256 as _keccak
                    from circuits_halo2_amd.utils import ints_to_fr as _itf
                    gold = os.path.join(ROOT, "tests", "golden")
                    rinv = pow(1 << 256, -1, _M.R)
                    _ints = lambda b: [int.from_bytes(bytes(b[i:i + 32]), "little") * rinv % _M.R for i in range(0, len(b), 32)]
                    tree = MerkleSumTree.from_csv(os.path.join(gold, "entry_16.csv"), 2)
                    mp = tree.generate_proof(0)
                    asg = _M.reference_assignment(11, int.from_bytes(_keccak(mp["entry"][0].encode()), "big") % _M.R, [int(x) for x in mp["entry"][1]],
                                                  mp["path_indices"], _ints(mp["sibling_leaf_node_hash_preimage"]),
                                                  [_ints(p_) for p_ in mp["sibling_middle_node_hash_preimages"]])
                    p11 = sg.ParamsKZG.read(open(os.path.join(gold, "hermez-raw-11"), "rb"))
                    p11.precompute()
                    dv = lambda v: torch.from_numpy(_itf(v)).cuda()
                    fx, sgm, adv = [dv(c) for c in asg["fixed"]], [dv(c) for c in asg["sigma"]], [dv(c) for c in asg["advice"]]
                    pk11 = _pv.ProvingKey(p11, 11, fx, sgm)
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    pk11 = _pv.ProvingKey(p11, 11, fx, sgm)
                    kg_ms = (time.perf_counter() - t1) * 1e3
                    kat = json.load(open(os.path.join(gold, "kat.json")))
                    want = [(int(a, 16), int(b, 16)) for a, b in kat["fixed_comms"] + kat["permutation_comms"]]
                    pk11.vk_digest = int(kat["vk_digest"], 16)
                    _pv.create_proof(p11, pk11, adv, asg["instances"])
                    t1 = time.perf_counter()
                    _pv.create_proof(p11, pk11, adv, asg["instances"])
                    py_ms = (time.perf_counter() - t1) * 1e3
                    ref = {"keygen_ms": kg_ms, "create_proof_ms": py_ms, "rows_used": asg["rows_used"],
                           "verifying_key_matches_reference": pk11.fixed_comms + pk11.permutation_comms == want,
                           "public_inputs_match_reference_K5": [hex(v) for v in asg["instances"]][:2] == [kat["k5"]["leaf0"], kat["k5"]["root"]]}
                    exe11 = os.path.join(ROOT, "tools", "create_proof_cpp")
                    if os.path.exists(exe11) and not ("rocprof" in os.environ.get("LD_PRELOAD", "") or any(k_.startswith(("ROCPROF", "ROCP_")) for k_ in os.environ)):
                        with tempfile.TemporaryDirectory() as td:
                            _pv.export_bundle(os.path.join(td, "b.bin"), p11, pk11, adv, asg["instances"])
                            r = subprocess.run([exe11, os.path.join(td, "b.bin"), os.path.join(td, "p.bin"), "8"], capture_output=True, text=True, timeout=300)
                            if r.returncode == 0:
                                ref["create_proof_ms_cpp_driver"] = json.loads(r.stdout.strip().splitlines()[-1])["create_proof_ms"]
                    p11.free()
                    ref["note"] = ("MstInclusionCircuit<4,2,8> in the reference's own floor plan (mst_inclusion.reference_assignment), reference SRS, "
                                   "csv/entry_16.csv user 0; proofs of this kind are accepted by the reference's verifier contract (tests/test_verifier_cpu.py)")
                    line["reference_circuit_k11"] = ref
                except Exception as ex:
                    line["reference_circuit_k11"] = {"error": repr(ex)}
                # the reference criterion bench's shape (zk_prover/benches/full_solvency_flow.rs: LEVELS = 20, k = 13): tree of
                # 2^20 users on the device, inclusion witness of one user, key generation, proof (tools/full_flow.py)
                try:
                    from full_flow import run as run_full_flow
                    torch.cuda.empty_cache()
                    line["full_flow_levels20_k13"] = dict(run_full_flow(20, 13, cpp=not profiled17, nc=1),
                                                          note="the reference's criterion bench (benches/full_solvency_flow.rs: LEVELS = 20, N_CURRENCIES = 1, "
                                                               "N_BYTES = 8, k = 13), its six entries on the device: mst_build_ms / mst_build_sorted_ms (2^20 "
                                                               "synthetic users), keygen_vk_ms (17 commitments) / keygen_pk_ms (coefficient and coset forms), "
                                                               "full_prover_ms / full_verifier_ms (Blake2b flavour, Python driver, product verifier); plus the "
                                                               "inclusion witness in the reference circuit's floor plan (host, Python integers) and the "
                                                               "Keccak-flavour create_proof (Python driver / C++ driver)")
                except Exception as ex:
                    line["full_flow_levels20_k13"] = {"error": repr(ex)}
            except Exception as ex:
                line["extras_error"] = repr(ex)

        # ---- the host-pointer entry points: what a [patch] of best_multiexp / best_fft in an unmodified halo2 calls
        # (INTEGRATION.md section 2).  PCIe-inclusive, from ordinary pageable memory, in place through ctypes; never `value`
        if not args.no_extras and not args.strong_only and world == 1 and args.log_n >= 20:
            try:
                line["host_pointer_entry_points"] = host_pointer_extra(scal, bases, result)
            except Exception as ex:
                line["host_pointer_entry_points"] = {"error": repr(ex)}
        # ---- CPU baseline for the NTT numbers above (same oracle, same box)
        if not args.no_cpu and world == 1 and "ntt" in line:
            from oracle import oracle as O
            a = O.random_fr(7, 1 << 22)
            for cores in sorted({O.ncpu(), min(O.ncpu(), 16)}, reverse=True):
                t1 = time.perf_counter()
                O.best_fft(a, O.omega(22), 22, cores)
                cdt = time.perf_counter() - t1
                key = "cpu_baseline_2^22" if cores == O.ncpu() else "cpu_baseline_2^22_16_threads"
                line["ntt"][key] = {"ms": cdt * 1e3, "elements_per_s": (1 << 22) / cdt, "cores": cores,
                                    "affinity_cores": O.ncpu(), "nproc": os.cpu_count(), "kind": "port"}
        # ---- CPU baseline: the oracle's restated halo2 best_multiexp on this box's cores
        if not args.no_cpu and world == 1:
            from oracle import oracle as O
            # threads = every host core this process is allowed to run on (sched_getaffinity: the measured share of the box,
            # reported with nproc beside it); halo2's best_multiexp splits the points into one chunk per thread, so the thread
            # count changes the algorithm's window size too -- the 16-thread figure of the earlier rounds is kept beside it
            hs, hb = scal.cpu().numpy(), bases.cpu().numpy()
            entries = []
            for cores in sorted({O.ncpu(), min(O.ncpu(), 16)}, reverse=True):
                t1 = time.perf_counter()
                ref = O.best_multiexp(hs, hb, cores)
                cdt = time.perf_counter() - t1
                ok = bool((ref == result).all()) if world == 1 else None
                entries.append({"value": n / cdt, "unit": "points/s", "cores": cores, "affinity_cores": O.ncpu(), "nproc": os.cpu_count(),
                                "kind": "port",
                                "sample": f"one full 2^{args.log_n} MSM on the same inputs ({cdt:.2f} s), "
                                          f"halo2-shaped per-thread-chunked Pippenger (oracle/bn254_oracle.c)",
                                "matches_gpu_result": ok})
            # the baseline is the CPU's best showing: the faster of "one thread per core this process may use" and the 16
            # threads of the earlier rounds (on a box whose affinity mask is wider than its CPU quota the former oversubscribes)
            entries.sort(key=lambda e: -e["value"])
            line["cpu_baseline"] = dict(entries[0])
            if len(entries) > 1:
                line["cpu_baseline"]["other_thread_count"] = entries[1]
        if strong_line is not None:
            line["msm_strong_scaling_2^23"] = strong_line
        if batch_line is not None:
            line["batch_k17"] = batch_line
            if "proofs_per_s" in batch_line:
                line["proofs_per_s"] = batch_line["proofs_per_s"]
                batch_line["roofline"] = proof_roofline(1e3 / batch_line["proofs_per_s"] * world,
                                                        "per proof of the batch: GPU-time per proof = n_gpus / proofs_per_s")
                under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k_.startswith(("ROCPROF", "ROCP_")) for k_ in os.environ)
                if world == 1 and not args.no_cpu_share_sweep and not args.no_extras and not under_profiler and batch_line.get("proofs_total", 0) >= 64:
                    # what one of 8 / 4 / 2 ranks sharing this host's CPUs would get (child runs under --cpu-share)
                    batch_line["cpu_share_rehearsal"] = cpu_share_rehearsal(args, batch_line["in_flight"], batch_line.get("host_cores_busy_per_gpu"))
        # BASELINE's metric is "proof-gen wall-clock + MSM-points/sec": the proof half where a reader of `config` / `roofline`
        # finds it without opening the extras
        cp = line.get("create_proof_k17") or {}
        proof_ms = cp.get("ms_cpp_driver") or cp.get("ms")
        if proof_ms:
            cp["roofline"] = proof_roofline(proof_ms, "one k = 17 proof, wall clock of the compiled driver")
            budget = committed_proof_budget()
            if budget:
                # counters under the proof: what bounds each of its kernels, and how far the measured wall clock is from the time its
                # vector ALUs need at the present instruction counts
                cp["roofline"]["counter_budget"] = budget
                cp["roofline"]["kernels"] = budget["kernels"]
                if budget.get("issue_floor_ms"):
                    cp["roofline"]["issue_floor_ms"] = budget["issue_floor_ms"]
                    cp["roofline"]["issue_floor_over_wall_clock"] = budget["issue_floor_ms"] / proof_ms
            line["roofline"]["create_proof_k17"] = cp["roofline"]
        line["config"]["proof_gen_k17_ms"] = proof_ms
        line["config"]["proof_gen_k17_verified"] = cp.get("verified")
        line["config"]["proofs_per_s_1024"] = (batch_line or {}).get("proofs_per_s")
        line["config"]["proofs_per_s_1024_errors"] = (batch_line or {}).get("errors")
        line["config"]["sequential_ms_per_step"] = line["sequential"]["ms_per_step"]
        dump_acc_log("the whole run")      # (a profile of this process holds every launch up to here)
        if batch_line and "roofline" in batch_line:
            bb = committed_batch_budget()
            if bb and bb.get("issue_floor_ms_per_proof") and batch_line.get("proofs_per_s"):
                # the profile's instruction count priced per proof, against THIS run's rate
                bb["efficiency_at_this_runs_rate"] = bb["issue_floor_ms_per_proof"] * batch_line["proofs_per_s"] / 1e3
                batch_line["roofline"]["counter_budget"] = bb
            line["roofline"]["batch_k17_per_proof"] = batch_line["roofline"]
        fs.emit(line)
    return line


def main():
    try:
        return _main()
    except SystemExit:
        raise
    except BaseException as ex:     # noqa: BLE001 -- whatever it was, the run still owes its one JSON line
        import traceback
        traceback.print_exc()
        if _FS is not None:
            if _FS.base_line is not None and _FS.args.gpus == 1:      # the headline was measured: an extra threw
                _FS.extra_failed(f"{type(ex).__name__} in an extra: {ex}")
            _FS.fail(f"{type(ex).__name__}: {ex}", 1)
        raise


if __name__ == "__main__":
    main()
