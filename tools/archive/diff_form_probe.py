"""difference-form commitments (sg_commit basis 2) against plain fixed-base commit_lagrange at k = 17: a z-like column
(distinct values on the first rows, one value over the unused rows, blinding rows at the end), the grand-product group
of a proof (three such columns + one dense coefficient-form polynomial) and dense random columns (no gain expected)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.utils import random_fr_canonical
from circuits_halo2_amd.arithmetic import fr_to_montgomery

ffi.check(ffi.lib().sg_init(0))
k = 17
n = 1 << k
tau = fr_to_montgomery(torch.from_numpy(random_fr_canonical(3, 1)).cuda()).cpu().numpy()
params = sg.ParamsKZG.setup(k, tau)
t0 = time.perf_counter(); params.precompute(0); params.precompute(1); t1 = time.perf_counter(); params.precompute(2); t2 = time.perf_counter()
print(f"precompute: bases 0+1 {(t1 - t0) * 1e3:.1f} ms, basis 2 (prefix sums + table) {(t2 - t1) * 1e3:.1f} ms")


def rnd(seed):
    return fr_to_montgomery(torch.from_numpy(random_fr_canonical(seed, n)).cuda())


def zlike(seed, used):
    c = rnd(seed).view(n, 32)
    c[used:n - 6] = c[used - 1]
    return c.reshape(-1).contiguous()


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


for used in (2000, 8000, 32000):
    z = [zlike(10 + j, used) for j in range(3)]
    dense = rnd(20)
    a = params.commit_batch_mixed(z + [dense], [1, 1, 1, 0]); b = params.commit_batch_mixed(z + [dense], [2, 2, 2, 0])
    assert (a == b).all()
    print(f"used rows {used:6d}: grand-product group (3 z-like + 1 dense): plain {timeit(lambda: params.commit_batch_mixed(z + [dense], [1, 1, 1, 0])):.3f} ms, "
          f"difference form {timeit(lambda: params.commit_batch_mixed(z + [dense], [2, 2, 2, 0])):.3f} ms; "
          f"one z-like column: plain {timeit(lambda: params.commit_batch(z[:1], lagrange=True)):.3f} ms, difference form {timeit(lambda: params.commit_batch(z[:1], lagrange=True, diff=True)):.3f} ms")
d = [rnd(30 + j) for j in range(4)]
assert (params.commit_batch(d, lagrange=True) == params.commit_batch(d, lagrange=True, diff=True)).all()
print(f"4 dense random columns: plain {timeit(lambda: params.commit_batch(d, lagrange=True)):.3f} ms, difference form {timeit(lambda: params.commit_batch(d, lagrange=True, diff=True)):.3f} ms")
