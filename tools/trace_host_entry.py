"""a few calls of sg_commit / sg_msm_g1 from pageable host memory (2^20), for a kernel + memory-copy trace of the host-pointer path"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.arithmetic import fr_to_montgomery, g1_fixed_base_mul
from circuits_halo2_amd.utils import random_fr_canonical
L = ffi.lib(); ffi.check(L.sg_init(0))
for kv in filter(None, os.environ.get("SG_PARAMS", "").split(",")):
    name, val = kv.split("="); ffi.check(L.sg_set_param(name.encode(), int(val)))
n = 1 << 20
scal = fr_to_montgomery(torch.from_numpy(random_fr_canonical(1, n)).cuda())
bases = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(2, n)).cuda()))
hs, hb = scal.cpu().numpy().copy(), bases.cpu().numpy().copy()
res = np.zeros(64, dtype=np.uint8)
which = sys.argv[1] if len(sys.argv) > 1 else "commit"
params = sg.ParamsKZG(20, hb, hb)
import time
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if which == "commit":
        ffi.check(L.sg_commit(C.c_uint64(params.handle()), C.c_int(0), ffi.ptr(hs), C.c_size_t(n), ffi.ptr(res)))
    else:
        ffi.check(L.sg_msm_g1(ffi.ptr(hs), ffi.ptr(hb), C.c_size_t(n), ffi.ptr(res)))
    print(which, i, round((time.perf_counter() - t0) * 1e3, 3), "ms", flush=True)
params.free()
