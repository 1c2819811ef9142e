"""per-phase HIP-event times of a single commit: generic vs fixed-base, k = 17 and 20; SG_PARAMS-style args name=value"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.utils import random_fr_canonical
from circuits_halo2_amd.arithmetic import g1_fixed_base_mul, fr_to_montgomery

ffi.check(ffi.lib().sg_init(0))
for a in sys.argv[1:]:
    name, v = a.split("=")
    ffi.check(ffi.lib().sg_set_param(name.encode(), C.c_int64(int(v))))
def timed(params, sc):
    tm = ffi.MsmTimings(); out = np.zeros(64, dtype=np.uint8)
    best = None
    for _ in range(5):
        t = time.perf_counter()
        ffi.check(ffi.lib().sg_commit_dev_timed(C.c_uint64(params.handle()), 0, ffi.dev_ptr(sc), C.c_size_t(sc.numel() // 32),
                                                ffi.current_stream_ptr(), ffi.ptr(out), C.byref(tm)))
        wall = (time.perf_counter() - t) * 1e3
        if best is None or wall < best[0]:
            best = (wall, tm.digits_ms, tm.sort_ms, tm.accumulate_ms, tm.reduce_ms, tm.total_ms, tm.window_bits, tm.tasks, tm.max_bucket)
    return "wall %.3f | digits %.3f sort %.3f acc %.3f reduce %.3f gpu-total %.3f | c=%d tasks=%d max_bucket=%d" % best
KS = [int(x) for x in os.environ.get('KS', '17,20').split(',')]
MODES = os.environ.get('MODES', 'generic,fixed').split(',')
for k in KS:
    n = 1 << k
    bases = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(11, n)).cuda()))
    g = bases.cpu().numpy()
    params = sg.ParamsKZG(k, g, g)
    sc = fr_to_montgomery(torch.from_numpy(random_fr_canonical(100, n)).cuda())
    if 'generic' in MODES:
        print(f"k={k} generic:", timed(params, sc), flush=True)
    if 'fixed' in MODES:
        params.precompute(0)
        print(f"k={k} fixed  :", timed(params, sc), flush=True)
    params.free()
