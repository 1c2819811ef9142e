import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import circuits_halo2_amd as sg
from circuits_halo2_amd.arithmetic import fr_to_montgomery, best_fft_batch
from circuits_halo2_amd.utils import random_fr_canonical
for lg, cnt in ((17, 9), (20, 9), (11, 9), (14, 9)):
    dom = sg.EvaluationDomain(2, lg)
    w = dom.get_omega()
    vs = [fr_to_montgomery(torch.from_numpy(random_fr_canonical(i, 1 << lg)).cuda()) for i in range(cnt)]
    for v in vs: sg.best_fft(v, w, lg)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        for v in vs: sg.best_fft(v, w, lg)
    torch.cuda.synchronize(); one = (time.perf_counter() - t) / 10
    best_fft_batch(vs, w, lg)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): best_fft_batch(vs, w, lg)
    torch.cuda.synchronize(); bat = (time.perf_counter() - t) / 10
    print("2^%d x %d: one-by-one %.1f us, batch %.1f us" % (lg, cnt, one * 1e6, bat * 1e6), flush=True)
