import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.arithmetic import fr_to_montgomery
from circuits_halo2_amd.utils import random_fr_canonical
L = sg.lib()
cfgs = [{"ntt.tile_log": 10}, {"ntt.tile_log": 9, "ntt.max_multi_log": 9, "ntt.threads": 256}, {"ntt.tile_log": 9, "ntt.max_multi_log": 9, "ntt.threads": 512},
        {"ntt.tile_log": 10, "ntt.max_multi_log": 9}, {"ntt.tile_log": 10, "ntt.max_multi_log": 9, "ntt.threads": 512},
        {"ntt.tile_log": 10, "ntt.max_multi_log": 8, "ntt.threads": 512}, {"ntt.tile_log": 9, "ntt.max_multi_log": 8, "ntt.threads": 256},
        {"ntt.tile_log": 9, "ntt.max_multi_log": 7, "ntt.threads": 256}, {"ntt.tile_log": 8, "ntt.max_multi_log": 8, "ntt.threads": 128},
        {"ntt.tile_log": 8, "ntt.max_multi_log": 7, "ntt.threads": 128}, {"ntt.tile_log": 10, "ntt.max_single_log": 10}]
base = {"ntt.threads": 1024, "ntt.tile_log": 11, "ntt.max_multi_log": 10, "ntt.max_single_log": 11}
bufs = {lg: fr_to_montgomery(torch.from_numpy(random_fr_canonical(lg, 1 << lg)).cuda()) for lg in (17, 20, 22)}
for cfg in cfgs:
    p = dict(base); p.update(cfg)
    for k, v in p.items():
        ffi.check(L.sg_set_param(k.encode(), v))
    res = []
    for lg, a in bufs.items():
        ms = C.c_float(0)
        rc = L.sg_time_ntt_dev(ffi.dev_ptr(a), C.c_uint32(lg), 20, C.byref(ms))
        res.append("2^%d %.1f us" % (lg, ms.value * 1e3) if rc == 0 else "2^%d ERR %s" % (lg, L.sg_last_error().decode()))
    print(cfg, " | ".join(res), flush=True)
