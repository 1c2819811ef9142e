import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import circuits_halo2_amd as sg
from circuits_halo2_amd.arithmetic import fr_to_montgomery, g1_fixed_base_mul
from circuits_halo2_amd.utils import random_fr_canonical
n = 1 << 20
scal = fr_to_montgomery(torch.from_numpy(random_fr_canonical(1, n)).cuda())
bases = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(2, n)).cuda()))
torch.cuda.synchronize()
for _ in range(2): sg.best_multiexp(scal, bases)
ts = []
for i in range(200):
    t = time.perf_counter(); sg.best_multiexp(scal, bases); ts.append((time.perf_counter() - t) * 1e3)
ts = np.array(ts)
print("mean %.3f median %.3f min %.3f p95 %.3f max %.3f" % (ts.mean(), np.median(ts), ts.min(), np.percentile(ts, 95), ts.max()))
print("outliers >2.5ms at", [(i, round(float(v), 2)) for i, v in enumerate(ts) if v > 2.5][:20])
