import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import circuits_halo2_amd as sg
from circuits_halo2_amd.arithmetic import fr_to_montgomery, g1_fixed_base_mul
from circuits_halo2_amd.utils import random_fr_canonical
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << lg
scal = fr_to_montgomery(torch.from_numpy(random_fr_canonical(1, n)).cuda())
bases = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(2, n)).cuda()))
torch.cuda.synchronize()
sg.best_multiexp_batch([(scal, bases)] * cnt)
for r in range(3):
    t = time.perf_counter(); sg.best_multiexp_batch([(scal, bases)] * cnt); print("batch ms total", (time.perf_counter() - t) * 1e3)
