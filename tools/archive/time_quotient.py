"""times the quotient-numerator kernels at the k = 17 proof shape (ext_k = 20)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from circuits_halo2_amd import ffi
from circuits_halo2_amd.arithmetic import quotient_permutation, quotient_lookup
from circuits_halo2_amd.utils import random_fr_canonical, to_montgomery_host

ffi.check(ffi.lib().sg_init(0))
k, ext_k = 17, 20
ne = 1 << ext_k
def r(seed):
    return torch.from_numpy(to_montgomery_host(random_fr_canonical(seed, ne))).cuda()
ncols, chunk = 6, 4
zs = [r(1), r(2)]; cols = [r(10 + i) for i in range(ncols)]; sig = [r(20 + i) for i in range(ncols)]
l0, ll, la, vals = r(30), r(31), r(32), r(33)
b = to_montgomery_host(random_fr_canonical(40, 1))
lk = [r(50 + i) for i in range(5)]
for name, fn, arrays in [
        ("permutation", lambda: quotient_permutation(vals, zs, cols, sig, chunk, l0, ll, la, b, b, b, k, ext_k, 6), 4 + 2 * 2 + 12 + 1),
        ("lookup", lambda: quotient_lookup(vals, *lk, l0, ll, la, b, b, b, k, ext_k), 10 + 1)]:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: {ms:.4f} ms per launch, {ne / ms / 1e6:.2f} G rows/s, ~{arrays * 32 * ne / ms / 1e6:.0f} GB/s algorithmic")
