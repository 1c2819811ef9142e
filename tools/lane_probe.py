"""do calls from two host threads overlap on the device?  times the same set of MSMs / NTTs from one thread and split
over two (each on its own torch stream), for several sizes; run with GPU_MAX_HW_QUEUES=4 / 8 to see the queue effect"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi, arithmetic as A
from circuits_halo2_amd.utils import random_fr_canonical

ffi.check(ffi.lib().sg_init(0))
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for lg, reps in ((14, 48), (17, 24), (20, 8)):
    n = 1 << lg
    scal = [A.fr_to_montgomery(torch.from_numpy(random_fr_canonical(10 + i, n)).cuda()) for i in range(2)]
    bases = A.g1_fixed_base_mul(A.fr_to_montgomery(torch.from_numpy(random_fr_canonical(99, n)).cuda()))
    torch.cuda.synchronize()

    def run(which, count):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(count):
                sg.best_multiexp(scal[which], bases)

    def timed(nthreads):
        ths = [threading.Thread(target=run, args=(i % 2, reps // nthreads)) for i in range(nthreads)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        return (time.perf_counter() - t0) * 1e3
    for nt in (1, 2, 3):
        timed(nt)
    res = {nt: min(timed(nt) for _ in range(3)) for nt in (1, 2, 3)}
    print(f"MSM 2^{lg} x {reps}: " + ", ".join(f"{nt} thread(s) {ms:.2f} ms" for nt, ms in res.items()))
