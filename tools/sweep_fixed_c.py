"""single and small-batch fixed-base commit latency vs table window width, small k"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.utils import random_fr_canonical
from circuits_halo2_amd.arithmetic import g1_fixed_base_mul, fr_to_montgomery
ffi.check(ffi.lib().sg_init(0))
for k in [int(x) for x in sys.argv[1:]] or [13]:
    n = 1 << k
    g = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(11, n)).cuda())).cpu().numpy()
    params = sg.ParamsKZG(k, g, g)
    scal = [fr_to_montgomery(torch.from_numpy(random_fr_canonical(100 + i, n)).cuda()) for i in range(5)]
    def timeit(fn, reps=20):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3
    print(f"k={k} generic: single {timeit(lambda: params.commit(scal[0])):.3f} ms, batch of 3 {timeit(lambda: params.commit_batch(scal[:3])):.3f}, of 5 {timeit(lambda: params.commit_batch(scal)):.3f}", flush=True)
    for c in range(max(4, k - 4), 17):
        params.precompute(0, window_bits=c)
        print(f"k={k} fixed c={c}: single {timeit(lambda: params.commit(scal[0])):.3f} ms, batch of 3 {timeit(lambda: params.commit_batch(scal[:3])):.3f}, of 5 {timeit(lambda: params.commit_batch(scal)):.3f}", flush=True)
    params.free()
