import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import snapshot_tree
from circuits_halo2_amd import api, batch as B, ffi
levels, nc, k = 20, 2, 17
params, pk, vk = B.setup_on_all_ranks(k, None, levels, nc)
tree = snapshot_tree(levels, nc)
users = [(7919 * i + 13) % (1 << levels) for i in range(64)]
mode = sys.argv[1]
if mode == "lanes8":
    ffi.check(ffi.lib().sg_set_param(b"lanes", 8))
for infl in ((4, 6, 8) if mode != "four" else (4,)):
    res = B.prove_batch(tree, users, params, pk, levels, in_flight=infl, combine=False)
    print(mode, infl, len(res.proofs), len(res.errors), flush=True)
print("done", flush=True)
