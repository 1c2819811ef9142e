"""known-answer check of a large MSM: bases s_i * G, result must be <k, s> * G (oracle for the scalar side)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.utils import random_fr_canonical, to_montgomery_host
from circuits_halo2_amd.arithmetic import g1_fixed_base_mul, fr_to_montgomery
from oracle import oracle as O
ffi.check(ffi.lib().sg_init(0))
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg
s_h = to_montgomery_host(random_fr_canonical(5, n)); k_h = to_montgomery_host(random_fr_canonical(6, n))
s_d, k_d = torch.from_numpy(s_h).cuda(), torch.from_numpy(k_h).cuda()
bases = g1_fixed_base_mul(s_d)
torch.cuda.synchronize()
t = time.perf_counter(); got = sg.best_multiexp(k_d, bases); dt = time.perf_counter() - t
t = time.perf_counter(); got = sg.best_multiexp(k_d, bases); dt = time.perf_counter() - t
want = O.fixed_base_mul(O.fr_dot(k_h, s_h), 1)
print(f"2^{lg}: {dt * 1e3:.2f} ms, {n / dt / 1e6:.0f} M points/s, known answer {'OK' if (got == want).all() else 'MISMATCH'}")
sys.exit(0 if (got == want).all() else 1)
