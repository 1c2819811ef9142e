"""randomised soak of the MSM paths against the oracle: sizes, batch shapes, fixed / generic, skewed scalars,
parameter toggles; stops after SECONDS (default 120).  Exit code 1 on the first mismatch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi
from oracle import oracle as O

ffi.check(ffi.lib().sg_init(0))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(time.time()))
K = 14
n_max = 1 << K
base_sc = O.random_fr(1, n_max)
bases = O.fixed_base_mul(base_sc, O.ncpu())
bases_l = O.fixed_base_mul(O.random_fr(2, n_max), O.ncpu())
params = sg.ParamsKZG(K, bases, bases_l)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
t0, rounds = time.time(), 0
while time.time() - t0 < budget:
    rounds += 1
    for name in ("msm.two_pass", "msm.quad", "msm.red2d"):
        ffi.check(ffi.lib().sg_set_param(name.encode(), C.c_int64(int(rng.integers(0, 3)))))
    # round 4: the fold pass in front of the 2-D line sums (on / off; a quad or a lane per bucket), the reduction's chunk, sleeping waits
    ffi.check(ffi.lib().sg_set_param(b"msm.red2d_prefold", C.c_int64(int(rng.integers(0, 2)))))
    ffi.check(ffi.lib().sg_set_param(b"msm.prefold_quad_buckets", C.c_int64(int(rng.choice([0, 1 << 12, 1 << 15, 1 << 18])))))
    ffi.check(ffi.lib().sg_set_param(b"msm.red2d_max_sets", C.c_int64(int(rng.choice([4, 6, 8])))))
    ffi.check(ffi.lib().sg_set_param(b"msm.log_red_chunk", C.c_int64(int(rng.choice([0, 0, 2, 3, 4])))))
    ffi.check(ffi.lib().sg_set_param(b"host.wait_sleep_us", C.c_int64(int(rng.choice([0, 0, 20])))))
    ffi.check(ffi.lib().sg_set_param(b"msm.acc_threads", C.c_int64(int(rng.choice([0, 64, 128, 256])))))
    if rounds % 3 == 1:
        params.free()
        params = sg.ParamsKZG(K, bases, bases_l)
        if rng.integers(2):
            params.precompute(window_bits=int(rng.choice([0, 5, 9, 13, 16])))
    n = int(rng.choice([1, 2, 63, 64, 65, 1000, 4096, 5000, n_max - 1, n_max]))
    m = int(rng.integers(1, 9))
    kind = int(rng.integers(4))
    cols = []
    for i in range(m):
        if kind == 0: sc = O.random_fr(int(rng.integers(1 << 30)), n)
        elif kind == 1: sc = np.tile(O.random_fr(int(rng.integers(1 << 30)), 1), n)                    # one scalar everywhere
        elif kind == 2: sc = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in rng.integers(0, 3, size=n)), dtype=np.uint8).copy()
        else:
            sc = O.random_fr(int(rng.integers(1 << 30)), n); sc[: 32 * (n // 2)] = 0                     # half zeros
        cols.append(sc)
    flags = [bool(rng.integers(2)) for _ in range(m)]
    sparse_hint = 16 if rng.integers(2) else 0          # SG_BASIS_SPARSE on every column: same commitments
    want = np.stack([O.best_multiexp(c, (bases_l if f else bases)[: 64 * n], O.ncpu()) for c, f in zip(cols, flags)])
    got = params.commit_batch_mixed([dev(c) for c in cols], [int(f) | sparse_hint for f in flags])
    if not (got == want).all():
        print("MISMATCH commit_batch_mixed", n, m, kind, flags); sys.exit(1)
    one = params.commit(dev(cols[0])) if not flags[0] else params.commit_lagrange(dev(cols[0]))
    if not (one == want[0]).all():
        print("MISMATCH commit", n, kind); sys.exit(1)
    g = sg.best_multiexp(cols[0], (bases_l if flags[0] else bases)[: 64 * n])
    if not (g == want[0]).all():
        print("MISMATCH best_multiexp", n, kind); sys.exit(1)
print(f"soak ok: {rounds} rounds in {time.time() - t0:.0f} s")
