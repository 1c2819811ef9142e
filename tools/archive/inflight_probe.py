"""Batch throughput against proofs in flight and library lanes, with and without the re-verification of every proof
(profiles/r03_sweeps/commit_combiner.txt, second table); run on the GPU box."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import snapshot_tree
from circuits_halo2_amd import api, batch as B, ffi
levels, nc, k = 20, 2, 17
params, pk, vk = B.setup_on_all_ranks(k, None, levels, nc)
tree = snapshot_tree(levels, nc)
users = [(7919 * i + 13) % (1 << levels) for i in range(384)]
def prove_only(c):
    inst = c.instances()[0]
    return api._create_proof(params, pk, c, [inst], "evm"), inst
for lanes in (4, 8):
    ffi.check(ffi.lib().sg_set_param(b"lanes", lanes))
    for infl in (4, 6, 8):
        for name, fn in (("prove+verify", None), ("prove only", prove_only)):
            B.prove_batch(tree, users[:2 * infl], params, pk, levels, in_flight=infl, prove=fn, combine=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = B.prove_batch(tree, users, params, pk, levels, in_flight=infl, prove=fn, combine=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"lanes {lanes} in_flight {infl} {name}: {len(res.proofs)} proofs {len(res.errors)} errors {len(users)/dt:.1f}/s", flush=True)
