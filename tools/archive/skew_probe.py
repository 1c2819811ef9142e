import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import circuits_halo2_amd as sg
from circuits_halo2_amd.arithmetic import fr_to_montgomery, g1_fixed_base_mul
from circuits_halo2_amd.utils import random_fr_canonical
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
n = 1 << k
bases = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(2, n)).cuda()))
rng = np.random.default_rng(1)
def canon_small(vals):
    a = np.zeros((n, 32), dtype=np.uint8)
    a[:, :8] = np.asarray(vals, dtype=np.uint64).view(np.uint8).reshape(n, 8)
    return fr_to_montgomery(torch.from_numpy(a.reshape(-1)).cuda())
cases = {
  "uniform": fr_to_montgomery(torch.from_numpy(random_fr_canonical(1, n)).cuda()),
  "bytes (range-check column)": canon_small(rng.integers(0, 256, n)),
  "sorted bytes (permuted lookup)": canon_small(np.sort(rng.integers(0, 256, n))),
  "selector 0/1": canon_small(rng.integers(0, 2, n)),
  "all ones": canon_small(np.ones(n)),
  "sparse 1% < 2^64": canon_small(np.where(rng.random(n) < 0.01, rng.integers(1, 1 << 62, n), 0)),
  "all zero": canon_small(np.zeros(n)),
}
torch.cuda.synchronize()
for name, s in cases.items():
    sg.best_multiexp(s, bases)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): r, tm = sg.best_multiexp(s, bases, timings=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print("%-32s %.3f ms  acc %.3f red %.3f tasks %d max_bucket %d" % (name, dt * 1e3, tm["accumulate_ms"], tm["reduce_ms"], tm["tasks"], tm["max_bucket"]), flush=True)
