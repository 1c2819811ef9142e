"""Where the host CPU of a proof batch goes: the k = 17 batch of bench.py (256 proofs, 48 in flight by default) on the first
N CPUs of the affinity mask, then user + system time per THREAD (/proc/self/task/*/stat), split into the batch's worker
threads, the other Python threads (main, witness producer) and the threads Python did not start (HIP runtime, library), and
per-section thread-CPU inside the workers (witness hand-over, create_proof, re-verification).
usage (GPU box): python tools/batch_cpu_profile.py [cores=2] [proofs=256] [in_flight=48] [ENV=VALUE ...]"""
import os
import sys
import threading
import time

cores = int(sys.argv[1]) if len(sys.argv) > 1 else 2
total = int(sys.argv[2]) if len(sys.argv) > 2 else 256
in_flight = int(sys.argv[3]) if len(sys.argv) > 3 else 48
for kv in sys.argv[4:]:                                # runtime knobs to try, e.g. ROC_SYSTEM_SCOPE_SIGNAL=0 (set before HIP starts)
    key, val = kv.split("=", 1)
    os.environ[key] = val
allowed = sorted(os.sched_getaffinity(0))
if cores > 0:
    os.sched_setaffinity(0, allowed[:cores])          # before anything touches the GPU: every later thread inherits it

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import circuits_halo2_amd as sg
from circuits_halo2_amd import api, batch as B, ffi, verifier as V

ffi.check(sg.lib().sg_init(0))
levels, k, nc = 20, 17, 2
params, pk, vk = B.setup_on_all_ranks(k, None, levels, nc)
params.precompute()
tree = bench.snapshot_tree(levels, nc)
torch.cuda.synchronize()
users = [(7919 * i + 13) % (1 << levels) for i in range(total)]

CLK = os.sysconf("SC_CLK_TCK")


def thread_cpu():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            with open(f"/proc/self/task/{tid}/stat") as f:
                s = f.read()
            comm = s[s.index("(") + 1:s.rindex(")")]
            rest = s[s.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / CLK, int(rest[11]) / CLK, int(rest[12]) / CLK)
        except (OSError, ValueError):
            pass
    return out


# per-section thread CPU inside the workers
sections = {}
sec_lock = threading.Lock()


def wrap(name, fn):
    def inner(*a, **kw):
        c0, w0 = time.thread_time(), time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            with sec_lock:
                c, w, n = sections.get(name, (0.0, 0.0, 0))
                sections[name] = (c + time.thread_time() - c0, w + time.perf_counter() - w0, n + 1)
    return inner


api._create_proof = wrap("create_proof (compiled driver, launches + waits + host tails)", api._create_proof)
V.verify_proof = wrap("verify_proof (re-verification: host pairing)", V.verify_proof)
api._advice_columns = wrap("  of which _advice_columns", api._advice_columns)
worker_tids = set()
orig_init = api.MstInclusionCircuit.init_from_tree.__func__


def init_from_tree(cls, *a, **kw):
    worker_tids.add(threading.get_native_id())
    return orig_init(cls, *a, **kw)


api.MstInclusionCircuit.init_from_tree = classmethod(wrap("MstInclusionCircuit.init_from_tree (witness hand-over)", init_from_tree))

B.prove_batch(tree, users[:96], params, pk, levels, flavour="evm", in_flight=in_flight)     # warm-up: lanes, pools, threads
torch.cuda.synchronize()
sections.clear()
before = thread_cpu()
allocs0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
t0 = time.perf_counter()
res = B.prove_batch(tree, users, params, pk, levels, flavour="evm", in_flight=in_flight)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
after = thread_cpu()
py_tids = {t.native_id for t in threading.enumerate()}
groups = {"batch worker threads": [0.0, 0.0, 0.0, 0], "other Python threads (main, witness producer)": [0.0, 0.0, 0.0, 0],
          "threads Python did not start (HIP runtime, library)": [0.0, 0.0, 0.0, 0]}
detail = []
for tid, (comm, tot, ut, st) in after.items():
    b = before.get(tid, (comm, 0.0, 0.0, 0.0))
    d, du, ds = tot - b[1], ut - b[2], st - b[3]
    g = "batch worker threads" if tid in worker_tids else "other Python threads (main, witness producer)" if tid in py_tids else \
        "threads Python did not start (HIP runtime, library)"
    groups[g][0] += d; groups[g][1] += du; groups[g][2] += ds; groups[g][3] += 1
    if g != "batch worker threads" and d > 0.02:
        detail.append((d, tid, comm, g))
n = len(res.proofs)
print(" ".join(sys.argv[4:]) or "(default runtime settings)")
print(f"{n} proofs ({len(res.errors)} errors) in {dt:.2f} s = {n / dt:.1f} proofs/s on {cores} cores, {in_flight} in flight")
all_cpu = sum(g[0] for g in groups.values())
print(f"CPU of all threads: {all_cpu:.2f} s = {1e3 * all_cpu / n:.2f} ms per proof, {all_cpu / dt:.2f} cores busy")
for name, (d, du, ds, cnt) in groups.items():
    print(f"  {name}: {cnt} threads, {1e3 * d / n:.2f} ms per proof (user {1e3 * du / n:.2f}, system {1e3 * ds / n:.2f})")
for d, tid, comm, g in sorted(detail, reverse=True)[:8]:
    print(f"    thread {tid} '{comm}': {1e3 * d / n:.2f} ms per proof")
print(f"device allocations by torch's caching allocator during the batch: {torch.cuda.memory_stats().get('num_device_alloc', 0) - allocs0}")
print("inside the workers (thread CPU / wall per call):")
for name, (c, w, cnt) in sections.items():
    print(f"  {name}: {1e3 * c / max(cnt, 1):.2f} ms CPU, {1e3 * w / max(cnt, 1):.2f} ms wall, {cnt} calls")
