"""runs the k=17 x16 fixed-base commit batch a few times (for rocprofv3 kernel traces)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.utils import random_fr_canonical
from circuits_halo2_amd.arithmetic import g1_fixed_base_mul, fr_to_montgomery
ffi.check(ffi.lib().sg_init(0))
for a in sys.argv[1:]:
    name, v = a.split("=")
    ffi.check(ffi.lib().sg_set_param(name.encode(), C.c_int64(int(v))))
k = int(os.environ.get("K", "17")); M = int(os.environ.get("M", "16")); n = 1 << k
g = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(11, n)).cuda())).cpu().numpy()
params = sg.ParamsKZG(k, g, g)
if os.environ.get("FIXED", "1") == "1":
    params.precompute(0)
scal = [fr_to_montgomery(torch.from_numpy(random_fr_canonical(100 + i, n)).cuda()) for i in range(M)]
params.commit_batch(scal); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3): params.commit_batch(scal)
torch.cuda.synchronize()
print(f"k={k} M={M}: {(time.perf_counter() - t) / 3 * 1e3:.3f} ms per batch")
