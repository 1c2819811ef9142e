"""1024 inclusion proofs at k = 17 with 8 / 12 / 16 / 24 proofs in flight (default batch path: combiner, witnesses ahead),
two repeats each; the last batch is checked by the oracle's verifier."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import snapshot_tree, oracle_vk
from circuits_halo2_amd import api, batch as B
levels, nc, k = 20, 2, 17
params, pk, vk = B.setup_on_all_ranks(k, None, levels, nc)
tree = snapshot_tree(levels, nc)
users = [(7919 * i + 13) % (1 << levels) for i in range(1024)]
for infl in (8, 12, 16, 24):
    B.prove_batch(tree, users[:3 * infl], params, pk, levels, in_flight=infl)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = B.prove_batch(tree, users, params, pk, levels, in_flight=infl)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"in_flight {infl}: {len(res.proofs)} proofs {len(res.errors)} errors {len(users)/dt:.1f}/s", flush=True)
from oracle import summa_verifier as SV
ovk = oracle_vk(params, vk)
print("oracle accepts:", all(SV.verify(p, i, ovk) for p, i in list(res.proofs.values())[:3]), all(res.proofs[u][1] == tree.public_inputs(u) for u in users[::97]))
