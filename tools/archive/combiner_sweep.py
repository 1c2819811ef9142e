"""Commit combiner sweep: 384 inclusion proofs at k = 17 (LEVELS = 20) with many proofs in flight, the combiner's target batch size,
deadline and fused-job capacity varied (profiles/r03_sweeps/commit_combiner.txt); run on the GPU box."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import snapshot_tree
from circuits_halo2_amd import api, batch as B, ffi
levels, nc, k = 20, 2, 17
params, pk, vk = B.setup_on_all_ranks(k, None, levels, nc)
tree = snapshot_tree(levels, nc)
users = [(7919 * i + 13) % (1 << levels) for i in range(384)]
setp = lambda name, v: ffi.check(ffi.lib().sg_set_param(name.encode(), v))
def stats():
    a, b = C.c_uint64(0), C.c_uint64(0)
    ffi.lib().sg_commit_combine_stats(C.byref(a), C.byref(b))
    return a.value, b.value
def run(infl, comb, label):
    B.prove_batch(tree, users[:2 * infl], params, pk, levels, in_flight=infl, combine=comb)
    j0, r0 = stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = B.prove_batch(tree, users, params, pk, levels, in_flight=infl, combine=comb)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    j1, r1 = stats()
    fus = (r1 - r0) / max(1, j1 - j0)
    print(f"{label}: in_flight {infl} combine {comb}: {len(res.proofs)} proofs {len(res.errors)} errors {len(users)/dt:.1f}/s fusion {fus:.2f}", flush=True)
setp("commit.combine_runners", 1)
for lfe in (25, 26, 27):
    setp("msm.log_fuse_entries", lfe)
    for infl, target in ((12, 12), (16, 16), (24, 24)):
        setp("commit.combine_target", target); setp("commit.combine_wait_us", 5000)
        run(infl, True, f"log_fuse_entries {lfe} target {target}")
