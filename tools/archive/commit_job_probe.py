"""fixed-base commitment jobs of a k = 17 proof (1 dense, 5 dense, 3 diff + 1 dense) under task lengths L = 2^log_seg"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
for seg in (1, 2, 4, 8, 16, 1, 8):
    ffi.check(ffi.lib().sg_set_param(b"msm.red2d_fold", seg))
    row = []
    for name, (cols, flags) in jobs.items():
        out = params.commit_batch_mixed(cols, flags)
        if name not in ref: ref[name] = out.copy()
        assert (out == ref[name]).all()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): params.commit_batch_mixed(cols, flags)
        torch.cuda.synchronize(); row.append(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
    print(f"red2d_fold {seg} | " + " | ".join(row), flush=True)
