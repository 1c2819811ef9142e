"""A-B of the NTT pass with two DIT stages per sweep ("ntt.radix4"): a lone transform (sg_time_ntt_dev), and the shapes a k = 17 proof
issues -- the batched inverse transform of 5 columns and the 25 coset blocks of a phase (sg_coeff_to_cosets_batch_dev).
usage (GPU box): python tools/ab_ntt_radix4.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import arithmetic as A, ffi
from circuits_halo2_amd.domain import EvaluationDomain
from circuits_halo2_amd.utils import random_fr_canonical

ffi.check(sg.lib().sg_init(0))
L = ffi.lib()
k = 17
n = 1 << k
dom = EvaluationDomain(6, k)
cols = [A.fr_to_montgomery(torch.from_numpy(random_fr_canonical(10 + i, n)).cuda()) for i in range(5)]


def timed(fn, reps=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for r4, tile in ((0, 10), (1, 10), (0, 10), (1, 10), (0, 10), (1, 10)):
    ffi.set_param("ntt.radix4", 1 if r4 else 2)       # 1: always, 2: never (0: by size, the default)
    ffi.set_param("ntt.big_tile_log", tile)
    line = [f"radix4 {r4} big_tile_log {tile}:"]
    for kk in (17, 20, 22):
        a = torch.randint(0, 255, (32 << kk,), dtype=torch.uint8, device="cuda")
        ms = C.c_float()
        ffi.check(L.sg_time_ntt_dev(ffi.dev_ptr(a), C.c_uint32(kk), C.c_int(20), C.byref(ms)))
        line.append(f"2^{kk} {ms.value * 1e3:7.1f} us")
    line.append(f"| 5 columns to coefficients {timed(lambda: A.best_fft_batch(cols, dom.get_omega_inv(), k, dom.ifft_divisor())):7.1f} us")
    line.append(f"| 5 columns to 25 coset blocks {timed(lambda: dom.coeff_to_cosets_batch(cols)):7.1f} us")
    print(" ".join(line), flush=True)
ffi.set_param("ntt.radix4", 0)
ffi.set_param("ntt.big_tile_log", 10)
