import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import snapshot_tree, oracle_vk
from circuits_halo2_amd import api, batch as B, verifier as V, prover as P
levels, nc, k = 20, 2, 17
params, pk, vk = B.setup_on_all_ranks(k, None, levels, nc)
tree = snapshot_tree(levels, nc)
users = [(7919 * i + 13) % (1 << levels) for i in range(1024)]
bad = {}
lock = threading.Lock()
def prove(c):
    inst = c.instances()[0]
    proof = api._create_proof(params, pk, c, [inst], "evm")
    ok1 = V.verify_proof(params, pk.vk, proof, inst, "evm")
    if not ok1:
        ok2 = V.verify_proof(params, pk.vk, proof, inst, "evm")
        try:
            ok3 = V._verify(params, pk.vk, proof, inst, "evm")
        except Exception as ex:
            ok3 = repr(ex)
        with lock:
            bad[len(bad)] = (ok1, ok2, ok3)
    return proof, inst
for infl in (1, 2, 4):
    bad.clear()
    t0 = time.perf_counter()
    res = B.prove_batch(tree, users[:512], params, pk, levels, in_flight=infl, prove=prove)
    dt = time.perf_counter() - t0
    print("in_flight", infl, "proofs", len(res.proofs), "errors", len(res.errors), list(res.errors.items())[:3], "bad", len(bad), list(bad.values())[:5], f"{512/dt:.1f}/s", flush=True)
from oracle import summa_verifier as SV
ovk = oracle_vk(params, vk)
if bad:
    print("oracle on first proofs:", [SV.verify(p, i, ovk) for p, i in list(res.proofs.values())[:3]])
