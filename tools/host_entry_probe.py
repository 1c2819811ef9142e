"""The host-pointer entry points of the C ABI (what a [patch] of best_multiexp / best_fft in an unmodified halo2 calls:
INTEGRATION.md section 2) timed through ctypes, in place, no Python-side copies: sg_msm_g1, sg_commit, sg_ntt_fr at 2^20
from pageable host memory, from page-locked memory and from pageable memory registered with sg_host_register; and
the raw transfer rates underneath (hipMemcpy through torch: pageable / pinned, both directions).
usage: python tools/host_entry_probe.py [json out]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.arithmetic import fr_to_montgomery, g1_fixed_base_mul
from circuits_halo2_amd.utils import random_fr_canonical

L = ffi.lib()
ffi.check(L.sg_init(0))
LOG = 20
n = 1 << LOG
scal = fr_to_montgomery(torch.from_numpy(random_fr_canonical(1, n)).cuda())
bases = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(2, n)).cuda()))
want = sg.best_multiexp(scal, bases)
w = sg.EvaluationDomain(2, LOG).get_omega()
ntt_want = sg.best_fft(scal.clone(), w, LOG).cpu().numpy()
out = {}


def timed(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]


def raw_rates():
    r = {}
    for name, nbytes in (("32MiB", 32 << 20), ("64MiB", 64 << 20)):
        page = np.ones(nbytes, dtype=np.uint8)
        pin = torch.ones(nbytes, dtype=torch.uint8).pin_memory()
        dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        tp = torch.from_numpy(page)

        def h2d(src):
            dev.copy_(src, non_blocking=True)
            torch.cuda.synchronize()

        def d2h(dst):
            dst.copy_(dev, non_blocking=True)
            torch.cuda.synchronize()
        r[name] = {"h2d_pageable_GBs": nbytes / timed(lambda: h2d(tp))[0] / 1e6, "h2d_pinned_GBs": nbytes / timed(lambda: h2d(pin))[0] / 1e6,
                   "d2h_pageable_GBs": nbytes / timed(lambda: d2h(tp))[0] / 1e6, "d2h_pinned_GBs": nbytes / timed(lambda: d2h(pin))[0] / 1e6}
    return r


out["raw"] = raw_rates()
print(json.dumps(out["raw"]), flush=True)


def entry_points(hs, hb, ha, label):
    res = np.zeros(64, dtype=np.uint8)
    p_s, p_b, p_a, p_r = (C.c_void_p(x.ctypes.data) for x in (hs, hb, ha, res))

    def msm():
        ffi.check(L.sg_msm_g1(p_s, p_b, C.c_size_t(n), p_r))
    row = {"sg_msm_g1_ms": timed(msm)}
    assert (res == want).all(), label
    params = sg.ParamsKZG(LOG, bases.cpu().numpy(), bases.cpu().numpy())
    try:
        def commit():
            ffi.check(L.sg_commit(C.c_uint64(params.handle()), C.c_int(0), p_s, C.c_size_t(n), p_r))
        row["sg_commit_ms"] = timed(commit)
        assert (res == want).all(), label
    finally:
        params.free()
    wv = ffi.u8(w)

    def ntt():
        ffi.check(L.sg_ntt_fr(p_a, ffi.ptr(wv), C.c_uint32(LOG)))
    src = scal.cpu().numpy()
    ha[:] = src
    ntt()
    assert (ha == ntt_want).all(), label
    row["sg_ntt_fr_ms"] = timed(ntt)            # (transforms its own output again and again: the time does not depend on the values)
    out[label] = row
    print(label, json.dumps(row), flush=True)


hs, hb = scal.cpu().numpy().copy(), bases.cpu().numpy().copy()
ha = hs.copy()
entry_points(hs, hb, ha, "pageable")

if hasattr(L, "sg_host_register"):
    t0 = time.perf_counter()
    for a in (hs, hb, ha):
        ffi.check(L.sg_host_register(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)))
    out["register_128MiB_ms"] = (time.perf_counter() - t0) * 1e3
    entry_points(hs, hb, ha, "registered")
    for a in (hs, hb, ha):
        ffi.check(L.sg_host_unregister(C.c_void_p(a.ctypes.data)))

ps, pb = torch.from_numpy(hs).pin_memory(), torch.from_numpy(hb).pin_memory()
pa = torch.from_numpy(ha).pin_memory()
entry_points(ps.numpy(), pb.numpy(), pa.numpy(), "page_locked")
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
