import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import circuits_halo2_amd as sg
from circuits_halo2_amd.arithmetic import fr_to_montgomery, g1_fixed_base_mul
from circuits_halo2_amd.utils import random_fr_canonical
n = 1 << 20
scal = fr_to_montgomery(torch.from_numpy(random_fr_canonical(1, n)).cuda())
bases = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(2, n)).cuda()))
hs, hb = scal.cpu().numpy(), bases.cpu().numpy()
for _ in range(2): sg.best_multiexp(hs, hb)
t = time.perf_counter()
for _ in range(5): r = sg.best_multiexp(hs, hb)
dt = (time.perf_counter() - t) / 5
print("host-buffer MSM 2^20 (pageable numpy, H2D included): %.3f ms -> %.1f M points/s" % (dt * 1e3, n / dt / 1e6))
params = sg.ParamsKZG(20, hb, hb)
for _ in range(2): params.commit(hs)
t = time.perf_counter()
for _ in range(5): r2 = params.commit(hs)
dt = (time.perf_counter() - t) / 5
print("host scalars, SRS resident (sg_commit): %.3f ms -> %.1f M points/s" % (dt * 1e3, n / dt / 1e6))
a = hs[: 32 << 20].copy()
w = sg.EvaluationDomain(2, 20).get_omega()
for _ in range(2): sg.best_fft(a, w, 20)
t = time.perf_counter()
for _ in range(5): sg.best_fft(a, w, 20)
dt = (time.perf_counter() - t) / 5
print("host-buffer NTT 2^20 (H2D + D2H included): %.3f ms" % (dt * 1e3))
