import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.arithmetic import fr_to_montgomery, g1_fixed_base_mul
from circuits_halo2_amd.utils import random_fr_canonical
from concurrent.futures import ThreadPoolExecutor
ffi.check(sg.lib().sg_init(0))
for lg in (20, 19, 22):
    n = 1 << lg
    sc = fr_to_montgomery(torch.from_numpy(random_fr_canonical(1, n)).cuda())
    bases = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(2, n)).cuda()))
    ref = None
    for rep in range(2):
        for groups in (0, 1):
            ffi.check(sg.lib().sg_set_param(b"msm.window_groups", groups))
            for _ in range(3):
                out = sg.best_multiexp(sc, bases)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30):
                out = sg.best_multiexp(sc, bases)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
            if ref is None: ref = out
            assert (out == ref).all()
            tls = __import__("threading").local()
            def one(_):
                if not hasattr(tls, "s"): tls.s = torch.cuda.Stream()
                with torch.cuda.stream(tls.s):
                    return sg.best_multiexp(sc, bases)
            with ThreadPoolExecutor(3) as tp:
                list(tp.map(one, range(6)))
                torch.cuda.synchronize(); t0 = time.perf_counter()
                outs = list(tp.map(one, range(45)))
                torch.cuda.synchronize(); dt3 = (time.perf_counter() - t0) / 45
            assert all((o == ref).all() for o in outs)
            print(f"2^{lg} window_groups {groups}: sequential {dt*1e3:.3f} ms, three in flight {dt3*1e3:.3f} ms per MSM", flush=True)
