"""latency of sg_fr_batch_invert_dev (one field inversion per thread: f29_inv_safegcd) at several sizes"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from circuits_halo2_amd import ffi, arithmetic as A
ffi.check(ffi.lib().sg_init(0))
for n in (8, 2048, 131072, 1 << 20):
    a = A.fr_random(bytes(32), 1, n)
    A.batch_invert(a.clone()); torch.cuda.synchronize()
    best = 1e9
    for _ in range(10):
        b = a.clone(); torch.cuda.synchronize(); t = time.perf_counter()
        A.batch_invert(b); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"batch_invert n={n}: {best*1e6:.1f} us")
