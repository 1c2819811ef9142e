"""One batch of inclusion proofs at k = 17 for traces: run_batch.py <in flight> <proofs> <combine 0|1>
(rocprofv3 --kernel-trace -- python3 tools/run_batch.py 16 256 1, then tools/batch_concurrency.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import snapshot_tree
from circuits_halo2_amd import batch as B
levels, nc, k = 20, 2, 17
infl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
count = int(sys.argv[2]) if len(sys.argv) > 2 else 96
comb = len(sys.argv) > 3 and sys.argv[3] == "1"
from circuits_halo2_amd import ffi
ffi.check(ffi.lib().sg_init(0))
for kv in filter(None, os.environ.get("SG_PARAMS", "").split(",")):   # tuning sweeps, as bench.py takes them
    name, val = kv.split("=")
    ffi.check(ffi.lib().sg_set_param(name.encode(), int(val)))
params, pk, vk = B.setup_on_all_ranks(k, None, levels, nc)
tree = snapshot_tree(levels, nc)
users = [(7919 * i + 13) % (1 << levels) for i in range(count)]
B.prove_batch(tree, users[:3 * infl], params, pk, levels, in_flight=infl, combine=comb)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = B.prove_batch(tree, users, params, pk, levels, in_flight=infl, combine=comb)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"in_flight {infl} combine {comb}: {len(res.proofs)} proofs, {len(res.errors)} errors, {count / dt:.1f}/s", flush=True)

