"""why does the generic fused 16 x 2^17 job take 7-15 ms inside bench.py and 5.5 ms alone?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.utils import random_fr_canonical
from circuits_halo2_amd.arithmetic import g1_fixed_base_mul, fr_to_montgomery
ffi.check(ffi.lib().sg_init(0))
n = 1 << 20
scal = fr_to_montgomery(torch.from_numpy(random_fr_canonical(1, n)).cuda())
bases = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(2, n)).cuda()))
k = 17
s17, b17 = scal[: 32 << k], bases[: 64 << k]
others = [fr_to_montgomery(torch.from_numpy(random_fr_canonical(10 + i, 1 << k)).cuda()) for i in range(16)]
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
print("fresh: same scalars x16      %.3f ms" % t(lambda: sg.best_multiexp_batch([(s17, b17)] * 16)))
print("fresh: distinct scalars x16  %.3f ms" % t(lambda: sg.best_multiexp_batch([(o, b17) for o in others])))
for _ in range(3): sg.best_multiexp(scal, bases)
sg.best_multiexp_batch([(scal, bases)] * 8)
print("after 2^20 work: same x16    %.3f ms" % t(lambda: sg.best_multiexp_batch([(s17, b17)] * 16)))
print("after 2^20 work: distinct    %.3f ms" % t(lambda: sg.best_multiexp_batch([(o, b17) for o in others])))
