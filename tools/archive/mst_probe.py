import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from circuits_halo2_amd.arithmetic import fr_to_montgomery
from circuits_halo2_amd.utils import random_fr_canonical
L = sg.lib()
for depth, nc in ((16, 2), (20, 1), (20, 2)):
    n = 1 << depth
    users = fr_to_montgomery(torch.from_numpy(random_fr_canonical(1, n)).cuda())
    bal = torch.zeros(32 * n * nc, dtype=torch.uint8, device="cuda")
    bal.view(n * nc, 32)[:, :7] = torch.randint(0, 256, (n * nc, 7), dtype=torch.uint8, device="cuda")
    bal = fr_to_montgomery(bal)
    h = torch.empty(32 * (2 * n - 1), dtype=torch.uint8, device="cuda")
    b = torch.empty(32 * (2 * n - 1) * nc, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        ffi.check(L.sg_mst_build_dev(ffi.dev_ptr(users), ffi.dev_ptr(bal), C.c_uint32(depth), C.c_uint32(nc), ffi.dev_ptr(h), ffi.dev_ptr(b), None))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        ffi.check(L.sg_mst_build_dev(ffi.dev_ptr(users), ffi.dev_ptr(bal), C.c_uint32(depth), C.c_uint32(nc), ffi.dev_ptr(h), ffi.dev_ptr(b), None))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    perms = n * (nc + 1) + (n - 1) * (nc + 2)
    print("depth %d nc %d: %.2f ms, %.1f M permutations/s, %.1f M leaves/s" % (depth, nc, dt * 1e3, perms / dt / 1e6, n / dt / 1e6), flush=True)
