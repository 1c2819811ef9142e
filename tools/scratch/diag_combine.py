import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import snapshot_tree, oracle_vk
from circuits_halo2_amd import api, batch as B, ffi
levels, nc, k = 20, 2, 17
params, pk, vk = B.setup_on_all_ranks(k, None, levels, nc)
tree = snapshot_tree(levels, nc)
users = [(7919 * i + 13) % (1 << levels) for i in range(512)]
def stats():
    a, b = C.c_uint64(0), C.c_uint64(0)
    ffi.lib().sg_commit_combine_stats(C.byref(a), C.byref(b))
    return a.value, b.value
B.prove_batch(tree, users[:16], params, pk, levels, in_flight=4, combine=False)
B.prove_batch(tree, users[:16], params, pk, levels, in_flight=4, combine=True)
for wait in (300, 100, 1000):
    ffi.check(ffi.lib().sg_set_param(b"commit.combine_wait_us", wait))
    for infl in (2, 3, 4, 5, 6):
        for comb in (False, True):
            j0, r0 = stats()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = B.prove_batch(tree, users, params, pk, levels, in_flight=infl, combine=comb)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            j1, r1 = stats()
            print(f"wait {wait} in_flight {infl} combine {comb}: {len(res.proofs)} proofs {len(res.errors)} errors {512/dt:.1f}/s fused jobs {j1-j0} requests {r1-r0}", flush=True)
        if wait != 300 and infl >= 4: break
from oracle import summa_verifier as SV
ovk = oracle_vk(params, vk)
print("oracle accepts:", all(SV.verify(p, i, ovk) for p, i in list(res.proofs.values())[:3]))
