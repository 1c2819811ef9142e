"""proofs per second of the k = 17 batch (two / three proofs in flight) under different library parameters: the
defaults are tuned for the latency of ONE proof (short accumulation tasks + quad-cooperative merges / reductions use
more lanes to shorten dependent chains); with several proofs in flight the chip is full and lane-efficient settings
may win"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench import snapshot_tree
from circuits_halo2_amd import api, batch as B, ffi

ffi.check(ffi.lib().sg_init(0))
levels, k, nc = 20, 17, 2
params, pk, vk = api.generate_setup_artifacts(k, None, api.MstInclusionCircuit.init_empty(levels, nc, 8))
params.precompute()
tree = snapshot_tree(levels, nc)
users = [(7919 * i + 13) % (1 << levels) for i in range(36)]
B.prove_batch(tree, users[:6], params, pk, levels, in_flight=3)
settings = [{}, {"msm.log_seg": 6}, {"msm.quad": 0}, {"msm.log_seg": 6, "msm.quad": 0}, {"msm.log_seg": 5, "msm.quad": 0},
            {"msm.log_seg": 6, "msm.quad": 0, "msm.red2d": 0}]
base = {"msm.log_seg": 0, "msm.quad": 1, "msm.red2d": 1}
for st in settings:
    for name, val in {**base, **st}.items():
        ffi.check(ffi.lib().sg_set_param(name.encode(), int(val)))
    B.prove_batch(tree, users[:6], params, pk, levels, in_flight=3)
    out = []
    for f in (1, 2, 3, 4):
        torch.cuda.synchronize()
        res = B.prove_batch(tree, users, params, pk, levels, in_flight=f)
        assert not res.errors, res.errors
        out.append(f"{f}: {res.proofs_per_s():6.1f}")
    print(st or "defaults", " | ".join(out), flush=True)
