"""fixed-base (precomputed window table) vs generic MSM: single and fused, k = 17 and k = 20"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.utils import random_fr_canonical, to_montgomery_host
from circuits_halo2_amd.arithmetic import g1_fixed_base_mul, fr_to_montgomery

ffi.check(ffi.lib().sg_init(0))
wb = [int(x) for x in sys.argv[1:]] or [0]
for k in (17, 20):
    n = 1 << k
    base_sc = fr_to_montgomery(torch.from_numpy(random_fr_canonical(11, n)).cuda())
    bases = g1_fixed_base_mul(base_sc)
    g = bases.cpu().numpy()
    params = sg.ParamsKZG(k, g, g)
    M = 16 if k == 17 else 4
    scal = [fr_to_montgomery(torch.from_numpy(random_fr_canonical(100 + i, n)).cuda()) for i in range(M)]
    def timeit(fn, reps=5):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3
    ref1 = params.commit(scal[0]); refb = params.commit_batch(scal)
    print(f"k={k} generic: single {timeit(lambda: params.commit(scal[0])):.3f} ms, batch of {M}: {timeit(lambda: params.commit_batch(scal)):.3f} ms", flush=True)
    for c in wb:
        t0 = time.perf_counter()
        params.precompute(0, window_bits=c)
        tp = (time.perf_counter() - t0) * 1e3
        ok = (params.commit(scal[0]) == ref1).all() and (params.commit_batch(scal) == refb).all()
        print(f"k={k} fixed c={c}: precompute {tp:.1f} ms, single {timeit(lambda: params.commit(scal[0])):.3f} ms, "
              f"batch of {M}: {timeit(lambda: params.commit_batch(scal)):.3f} ms, same bits: {ok}", flush=True)
    params.free()
