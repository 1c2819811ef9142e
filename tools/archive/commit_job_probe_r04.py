"""fixed-base commitment jobs of a k = 17 proof (1 dense, 5 dense, 3 z-like + 1 dense), alternating msm.red2d_prefold
= 1 (partial sums folded once by msm_fold_buckets, line sums over plain arrays) and 0 (the line sums fold them, twice)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.utils import random_fr_canonical
from circuits_halo2_amd.arithmetic import g1_fixed_base_mul, fr_to_montgomery
ffi.check(ffi.lib().sg_init(0))
k = 17; n = 1 << k
bases = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(11, n)).cuda())).cpu().numpy()
params = sg.ParamsKZG(k, bases, bases); params.precompute()
dense = [fr_to_montgomery(torch.from_numpy(random_fr_canonical(100 + i, n)).cuda()) for i in range(5)]
zlike = []
for i in range(3):
    z = dense[i].clone().view(-1, 32)
    z[9000:] = z[9000]
    zlike.append(z.reshape(-1).contiguous())
jobs = {"1 dense": (dense[:1], [0]), "5 dense": (dense, [0] * 5), "3 z-like + 1 dense": (zlike + dense[:1], [2, 2, 2, 0])}
ref = {}
settings = [("msm.red2d_prefold", 1), ("msm.red2d_prefold", 0), ("msm.red2d_prefold", 1), ("msm.red2d_prefold", 0),
            ("msm.prefold_quad_buckets", 0), ("msm.prefold_quad_buckets", 1 << 18), ("msm.prefold_quad_buckets", 1 << 15)]
for name, val in settings:
    ffi.check(ffi.lib().sg_set_param(b"msm.red2d_prefold", 1))
    ffi.check(ffi.lib().sg_set_param(name.encode(), val))
    row = []
    for jn, (cols, flags) in jobs.items():
        out = params.commit_batch_mixed(cols, flags)
        if jn not in ref: ref[jn] = out.copy()
        assert (out == ref[jn]).all()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): params.commit_batch_mixed(cols, flags)
        torch.cuda.synchronize(); row.append(f"{jn}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
    print(f"{name} {val} | " + " | ".join(row), flush=True)
