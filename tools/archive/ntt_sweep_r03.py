"""Re-sweep of the NTT plan parameters with the round-2 closings in place (zero-padded first passes, limb-step closings,
column-chain products): standalone 2^17 and 2^22 (sg_time_ntt_dev), a batch of 16 transforms of 2^17 and the 9-column
coset transform of a k = 17 proof, for every (per-pass length, tile, threads) combination that is legal.
usage (GPU box): python tools/ntt_sweep_r03.py > gpurun_out/ntt_sweep_r03.txt"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.arithmetic import fr_to_montgomery
from circuits_halo2_amd.domain import EvaluationDomain
from circuits_halo2_amd.utils import random_fr_canonical

L = sg.lib()
k = 17
bufs = {lg: fr_to_montgomery(torch.from_numpy(random_fr_canonical(lg, 1 << lg)).cuda()) for lg in (17, 22)}
batch = [fr_to_montgomery(torch.from_numpy(random_fr_canonical(100 + i, 1 << k)).cuda()) for i in range(16)]
dom = EvaluationDomain(6, k)
omega = ffi.u8(dom.get_omega())
cos_out = [torch.empty(32 * 5 << k, dtype=torch.uint8, device="cuda") for _ in range(9)]


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def batch16():
    ptrs = (C.c_void_p * 16)(*[b.data_ptr() for b in batch])
    ffi.check(L.sg_ntt_fr_batch_dev(ptrs, C.c_size_t(16), ffi.ptr(omega), None, C.c_uint32(k), ffi.current_stream_ptr()))


def cosets9():
    ins = (C.c_void_p * 9)(*[b.data_ptr() for b in batch[:9]])
    outs = (C.c_void_p * 9)(*[b.data_ptr() for b in cos_out])
    ffi.check(L.sg_coeff_to_cosets_batch_dev(ins, outs, C.c_size_t(9), C.c_uint32(k), C.c_uint32(k + 3), C.c_uint32(5), ffi.current_stream_ptr()))


print("config (max_multi_log, tile_log, threads) | 2^17 us | 2^22 us | 16 x 2^17 batch us | 9-column coset transform (45 x 2^17) us")
for multi in (8, 9):
    for tile in (8, 9, 10, 11, 12):
        for threads in (128, 256, 512, 1024):
            if tile < multi or threads > (1 << tile) // 2 or (tile >= 11 and threads < 256):
                continue
            for name, v in (("ntt.big_tile_log", 0), ("ntt.max_single_log", 11), ("ntt.max_multi_log", multi), ("ntt.tile_log", tile), ("ntt.threads", threads)):
                ffi.check(L.sg_set_param(name.encode(), v))
            res = []
            for lg in (17, 22):
                ms = C.c_float(0)
                rc = L.sg_time_ntt_dev(ffi.dev_ptr(bufs[lg]), C.c_uint32(lg), 20, C.byref(ms))
                res.append("%8.1f" % (ms.value * 1e3) if rc == 0 else "     ERR")
            try:
                res.append("%8.1f" % timed(batch16))
                res.append("%8.1f" % timed(cosets9))
            except Exception as ex:
                res.append(repr(ex)[:60])
            print(f"({multi}, {tile:2d}, {threads:4d}) | " + " | ".join(res), flush=True)

# the shipped defaults: latency shape for lone small transforms, throughput shape for batches and large transforms
for name, v in (("ntt.max_multi_log", 9), ("ntt.tile_log", 9), ("ntt.threads", 256), ("ntt.big_tile_log", 10), ("ntt.big_threads", 512)):
    ffi.check(L.sg_set_param(name.encode(), v))
for batch_min, big_log in ((4, 20), (2, 20), (8, 20), (4, 18), (4, 23)):
    ffi.check(L.sg_set_param(b"ntt.batch_min", batch_min)); ffi.check(L.sg_set_param(b"ntt.big_log", big_log))
    res = []
    for lg in (17, 22):
        ms = C.c_float(0)
        L.sg_time_ntt_dev(ffi.dev_ptr(bufs[lg]), C.c_uint32(lg), 20, C.byref(ms))
        res.append("%8.1f" % (ms.value * 1e3))
    res.append("%8.1f" % timed(batch16)); res.append("%8.1f" % timed(cosets9))
    print(f"auto shape: batch_min {batch_min} big_log {big_log} | " + " | ".join(res), flush=True)
