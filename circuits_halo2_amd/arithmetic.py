"""Mirror of `halo2_proofs::arithmetic` for the two functions on the hot path.

best_multiexp(coeffs, bases) -> G1   and   best_fft(a, omega, log_n)  [UPSTREAM crate absent
from /root/reference; reached from zk_prover/src/circuits/utils.rs:75,76,94-101].  Same
argument meaning and error behaviour: length mismatch is an assertion failure upstream and a
ValueError here; results are returned in halo2curves' byte layout (numpy uint8).

Each function accepts either host buffers (numpy uint8 / bytes) or device-resident torch
uint8 CUDA tensors; device inputs are consumed in place with no PCIe traffic.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import ffi


def _is_torch_cuda(x) -> bool:
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def best_multiexp(coeffs, bases, timings: bool = False):
    """sum_i coeffs[i] * bases[i]; coeffs n x 32 B Fr (Montgomery), bases n x 64 B G1Affine.
    Returns the 64-byte affine point (identity = zeros)."""
    L = ffi.lib()
    out = np.zeros(64, dtype=np.uint8)
    if _is_torch_cuda(coeffs):
        n = coeffs.numel() // 32
        if bases.numel() != 64 * n:
            raise ValueError("best_multiexp: coeffs.len() != bases.len()")
        if timings:
            tm = ffi.MsmTimings()
            ffi.check(L.sg_msm_g1_dev_timed(ffi.dev_ptr(coeffs), ffi.dev_ptr(bases), C.c_size_t(n),
                                            ffi.current_stream_ptr(), ffi.ptr(out), C.byref(tm)))
            return out, tm.as_dict()
        ffi.check(L.sg_msm_g1_dev(ffi.dev_ptr(coeffs), ffi.dev_ptr(bases), C.c_size_t(n), ffi.current_stream_ptr(),
                                  ffi.ptr(out)))
        return out
    s, b = ffi.u8(coeffs), ffi.u8(bases)
    if s.size % 32 or b.size % 64 or s.size // 32 != b.size // 64:
        raise ValueError("best_multiexp: coeffs.len() != bases.len()")
    ffi.check(L.sg_msm_g1(ffi.ptr(s), ffi.ptr(b), C.c_size_t(s.size // 32), ffi.ptr(out)))
    return out


def best_multiexp_batch(pairs):
    """[(coeffs, bases), ...] -> list of 64-byte affine points.  The independent MSMs of one
    prover phase in one call; they are pipelined on the device (C ABI: sg_msm_g1_batch).
    All pairs must be of one kind: host buffers or device tensors."""
    L = ffi.lib()
    count = len(pairs)
    out = np.zeros(64 * count, dtype=np.uint8)
    if count == 0:
        return []
    ns = (C.c_size_t * count)()
    ps = (C.c_void_p * count)()
    pb = (C.c_void_p * count)()
    keep = []
    dev = _is_torch_cuda(pairs[0][0])
    for i, (s, b) in enumerate(pairs):
        if dev:
            n = s.numel() // 32
            if b.numel() != 64 * n:
                raise ValueError("best_multiexp: coeffs.len() != bases.len()")
            ps[i], pb[i] = s.data_ptr(), b.data_ptr()
        else:
            s, b = ffi.u8(s), ffi.u8(b)
            n = s.size // 32
            if s.size % 32 or b.size != 64 * n:
                raise ValueError("best_multiexp: coeffs.len() != bases.len()")
            keep.append((s, b))
            ps[i], pb[i] = s.ctypes.data, b.ctypes.data
        ns[i] = n
    if dev:
        ffi.check(L.sg_msm_g1_batch_dev(ps, pb, ns, C.c_size_t(count), ffi.current_stream_ptr(), ffi.ptr(out)))
    else:
        ffi.check(L.sg_msm_g1_batch(ps, pb, ns, C.c_size_t(count), ffi.ptr(out)))
    return [out[64 * i:64 * i + 64].copy() for i in range(count)]


def best_fft(a, omega, log_n: int):
    """Forward DFT of 2^log_n Fr values with generator `omega`, natural order in and out.
    Host input: returns a new numpy buffer.  Device tensor: transformed in place (like the
    `&mut [Fr]` upstream) and returned."""
    L = ffi.lib()
    w = ffi.u8(omega)
    if w.size != 32:
        raise ValueError("best_fft: omega must be one 32-byte Fr")
    if _is_torch_cuda(a):
        if a.numel() != 32 << log_n:
            raise ValueError("best_fft: a.len() != 1 << log_n")
        ffi.check(L.sg_ntt_fr_dev(ffi.dev_ptr(a), ffi.ptr(w), C.c_uint32(log_n), ffi.current_stream_ptr()))
        return a
    buf = ffi.u8(a).copy()
    if buf.size != 32 << log_n:
        raise ValueError("best_fft: a.len() != 1 << log_n")
    ffi.check(L.sg_ntt_fr(ffi.ptr(buf), ffi.ptr(w), C.c_uint32(log_n)))
    return buf


def g1_fixed_base_mul(scalars):
    """out[i] = scalars[i] * G (what ParamsKZG::setup computes for g[]); host or device."""
    L = ffi.lib()
    if _is_torch_cuda(scalars):
        import torch
        n = scalars.numel() // 32
        out = torch.empty(64 * n, dtype=torch.uint8, device=scalars.device)
        ffi.check(L.sg_g1_fixed_base_mul_dev(ffi.dev_ptr(scalars), C.c_size_t(n), ffi.dev_ptr(out),
                                             ffi.current_stream_ptr()))
        return out
    s = ffi.u8(scalars)
    out = np.zeros(2 * s.size, dtype=np.uint8)
    ffi.check(L.sg_g1_fixed_base_mul(ffi.ptr(s), C.c_size_t(s.size // 32), ffi.ptr(out)))
    return out


def fr_to_montgomery(canon):
    """canonical 32-B LE integers (< r) -> Montgomery Fr, on the device (torch tensor in/out)."""
    import torch
    L = ffi.lib()
    out = torch.empty_like(canon)
    ffi.check(L.sg_fr_to_montgomery_dev(ffi.dev_ptr(canon), ffi.dev_ptr(out), C.c_size_t(canon.numel() // 32),
                                        ffi.current_stream_ptr()))
    return out


def fr_from_montgomery(mont):
    import torch
    L = ffi.lib()
    out = torch.empty_like(mont)
    ffi.check(L.sg_fr_from_montgomery_dev(ffi.dev_ptr(mont), ffi.dev_ptr(out), C.c_size_t(mont.numel() // 32),
                                          ffi.current_stream_ptr()))
    return out


def lookup_permute_small(inp, table, rows: int):
    """halo2's permute_expression_pair on the device for range tables (all table values < 2^16): returns the
    permuted input / table columns over `rows` usable rows (device tensors), or None when the table is outside
    that range (sort on the host instead); raises ValueError when an input value is not in the table"""
    import torch
    a = torch.empty(32 * rows, dtype=torch.uint8, device="cuda")
    s = torch.empty(32 * rows, dtype=torch.uint8, device="cuda")
    rc = ffi.lib().sg_lookup_permute_small_dev(ffi.dev_ptr(inp), ffi.dev_ptr(table), C.c_size_t(rows), ffi.dev_ptr(a), ffi.dev_ptr(s),
                                               ffi.current_stream_ptr())
    if rc == -5:
        return None
    if rc == -6:        # SG_ERR_WITNESS
        raise ValueError("lookup input value not in the table")
    ffi.check(rc)
    return a, s


def fr_random(key: bytes, stream_id: int, n: int):
    """n uniform field elements on the device from ChaCha20(key) (sg_fr_random_dev): blinding values of a proof"""
    import torch
    if len(key) != 32:
        raise ValueError("fr_random: 32-byte key")
    out = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    ffi.check(ffi.lib().sg_fr_random_dev(ffi.ptr(np.frombuffer(key, dtype=np.uint8).copy()), C.c_uint64(stream_id), ffi.dev_ptr(out),
                                         C.c_size_t(n), ffi.current_stream_ptr()))
    return out


# ---- helpers that keep vectors on the device between NTTs and MSMs ---------
def eval_polynomial(poly, point):
    """halo2_proofs::arithmetic::eval_polynomial: sum_i poly[i] * point^i -> 32-byte Fr"""
    L = ffi.lib()
    x = ffi.u8(point)
    out = np.zeros(32, dtype=np.uint8)
    if _is_torch_cuda(poly):
        ffi.check(L.sg_fr_eval_poly_dev(ffi.dev_ptr(poly), C.c_size_t(poly.numel() // 32), ffi.ptr(x),
                                        ffi.current_stream_ptr(), ffi.ptr(out)))
    else:
        c = ffi.u8(poly)
        ffi.check(L.sg_fr_eval_poly(ffi.ptr(c), C.c_size_t(c.size // 32), ffi.ptr(x), ffi.ptr(out)))
    return out


def batch_invert(a):
    """ff::BatchInvert on a device tensor, in place (zeros stay zero)"""
    ffi.check(ffi.lib().sg_fr_batch_invert_dev(ffi.dev_ptr(a), C.c_size_t(a.numel() // 32), ffi.current_stream_ptr()))
    return a


def prefix_product(a):
    """out[0] = 1, out[i] = a[0] * ... * a[i-1]; returns a device tensor of n + 1 elements"""
    import torch
    n = a.numel() // 32
    out = torch.empty(32 * (n + 1), dtype=torch.uint8, device=a.device)
    ffi.check(ffi.lib().sg_fr_prefix_product_dev(ffi.dev_ptr(a), C.c_size_t(n), ffi.dev_ptr(out),
                                                 ffi.current_stream_ptr()))
    return out


def fr_mul(a, b):
    import torch
    out = torch.empty_like(a)
    ffi.check(ffi.lib().sg_fr_mul_dev(ffi.dev_ptr(a), ffi.dev_ptr(b), C.c_size_t(a.numel() // 32), ffi.dev_ptr(out),
                                      ffi.current_stream_ptr()))
    return out


def eval_polynomial_batch(polys, points) -> np.ndarray:
    """polys[j](points[j]) for equal-length device polynomials; points: m x 32 bytes -> (m, 32) uint8"""
    m = len(polys)
    pts = ffi.u8(points)
    if pts.size != 32 * m:
        raise ValueError("eval_polynomial_batch: one point per polynomial")
    out = np.zeros((m, 32), dtype=np.uint8)
    if m == 0:
        return out
    n = polys[0].numel() // 32
    for p in polys:
        if p.numel() != 32 * n:
            raise ValueError("eval_polynomial_batch: equal lengths expected")
    ptrs = (C.c_void_p * m)(*[p.data_ptr() for p in polys])
    ffi.check(ffi.lib().sg_fr_eval_poly_batch_dev(ptrs, C.c_size_t(n), ffi.ptr(pts), C.c_uint32(m),
                                                  ffi.current_stream_ptr(), ffi.ptr(out)))
    return out


def kate_division(a, b, with_remainder: bool = False):
    """halo2 arithmetic::kate_division: quotient of a(X) by (X - b) on a device tensor; returns n - 1 coefficients
    (and a(b) when asked)"""
    import torch
    n = a.numel() // 32
    q = torch.empty(32 * n, dtype=torch.uint8, device=a.device)
    rem = np.zeros(32, dtype=np.uint8)
    ffi.check(ffi.lib().sg_fr_kate_division_dev(ffi.dev_ptr(a), C.c_size_t(n), ffi.ptr(ffi.u8(b)), ffi.dev_ptr(q),
                                                ffi.ptr(rem) if with_remainder else None, ffi.current_stream_ptr()))
    q = q[:32 * max(0, n - 1)]
    return (q, rem) if with_remainder else q


def kate_division_batch(polys, points):
    """exact quotients polys[j] / (X - points[j]) of equal-length device polynomials (m <= 16; a polynomial may appear
    several times), one launch per scan step for all of them: returns m device tensors of n coefficients (the last one 0)"""
    import torch
    m = len(polys)
    pts = ffi.u8(points)
    if pts.size != 32 * m or m > 16:
        raise ValueError("kate_division_batch: one point per polynomial, at most 16")
    if m == 0:
        return []
    n = polys[0].numel() // 32
    if any(p.numel() != 32 * n for p in polys):
        raise ValueError("kate_division_batch: equal lengths expected")
    outs = [torch.empty(32 * n, dtype=torch.uint8, device=polys[0].device) for _ in range(m)]
    pa = (C.c_void_p * m)(*[p.data_ptr() for p in polys])
    pq = (C.c_void_p * m)(*[q.data_ptr() for q in outs])
    ffi.check(ffi.lib().sg_fr_kate_division_batch_dev(pa, C.c_size_t(n), ffi.ptr(pts), C.c_uint32(m), pq, ffi.current_stream_ptr()))
    return outs


def count_noncanonical(cols):
    """device u32 tensor (1 element) = number of elements of the equal-length device columns whose word value is >= r;
    asynchronous on the current stream (read it at the next point where the host waits anyway)"""
    import torch
    m = len(cols)
    if m > 16:
        raise ValueError("count_noncanonical: at most 16 columns per call")
    n = cols[0].numel() // 32 if m else 0
    if any(c.numel() != 32 * n for c in cols):
        raise ValueError("count_noncanonical: equal lengths expected")
    count = torch.empty(1, dtype=torch.int32, device="cuda")
    pc = (C.c_void_p * max(m, 1))(*[c.data_ptr() for c in cols])
    ffi.check(ffi.lib().sg_fr_count_noncanonical_dev(pc, C.c_uint32(m), C.c_size_t(n), ffi.dev_ptr(count), ffi.current_stream_ptr()))
    return count


def lincomb(polys, coeffs):
    """sum_j coeffs[j] * polys[j] over equal-length device tensors; coeffs: m x 32 bytes"""
    import torch
    m = len(polys)
    c = ffi.u8(coeffs)
    if c.size != 32 * m:
        raise ValueError("lincomb: one coefficient per polynomial")
    for p in polys:
        if p.numel() != polys[0].numel():
            raise ValueError("lincomb: equal lengths expected")
    out = torch.empty_like(polys[0])
    ptrs = (C.c_void_p * m)(*[p.data_ptr() for p in polys])
    ffi.check(ffi.lib().sg_fr_lincomb_dev(ptrs, ffi.ptr(c), C.c_uint32(m), C.c_size_t(polys[0].numel() // 32),
                                          ffi.dev_ptr(out), ffi.current_stream_ptr()))
    return out


def permutation_product(values, sigmas, beta, gamma, delta_start, k: int, z0=None):
    """one chunk of halo2's permutation grand product on device tensors; returns z (2^k rows)"""
    import torch
    m = len(values)
    pv = (C.c_void_p * m)(*[v.data_ptr() for v in values])
    ps = (C.c_void_p * m)(*[s.data_ptr() for s in sigmas])
    z = torch.empty(32 << k, dtype=torch.uint8, device=values[0].device)
    z0p = ffi.ptr(ffi.u8(z0)) if z0 is not None else None
    ffi.check(ffi.lib().sg_permutation_product_dev(pv, ps, C.c_uint32(m), ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)),
                                                   ffi.ptr(ffi.u8(delta_start)), C.c_uint32(k), z0p, ffi.dev_ptr(z),
                                                   ffi.current_stream_ptr()))
    return z


def lookup_product(inp, table, permuted_input, permuted_table, beta, gamma):
    """halo2's lookup grand product on device tensors; returns z (n rows)"""
    import torch
    n = inp.numel() // 32
    z = torch.empty(32 * n, dtype=torch.uint8, device=inp.device)
    ffi.check(ffi.lib().sg_lookup_product_dev(ffi.dev_ptr(inp), ffi.dev_ptr(table), ffi.dev_ptr(permuted_input),
                                              ffi.dev_ptr(permuted_table), ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)),
                                              C.c_size_t(n), ffi.dev_ptr(z), ffi.current_stream_ptr()))
    return z


def grand_products(perm_chunks, lookups, beta, gamma, k: int, usable_rows: int):
    """ALL grand products of one proof in batched launches (sg_grand_products_dev): `perm_chunks` = [(values, sigmas), ...]
    in order (chunk j's z continues from chunk j-1's value at row `usable_rows`), `lookups` = [(input, table, permuted
    input, permuted table), ...]; returns ([z per chunk], [z per lookup]), 2^k rows each"""
    import torch
    vals = [v for ch in perm_chunks for v in ch[0]]
    sigs = [s_ for ch in perm_chunks for s_ in ch[1]]
    if len(vals) != len(sigs):
        raise ValueError("grand_products: one sigma per column")
    cols = (C.c_uint32 * max(1, len(perm_chunks)))(*[len(ch[0]) for ch in perm_chunks])
    pv = (C.c_void_p * max(1, len(vals)))(*[ffi.dev_ptr(v).value for v in vals])
    ps = (C.c_void_p * max(1, len(sigs)))(*[ffi.dev_ptr(v).value for v in sigs])
    lk = [t for lu in lookups for t in lu]
    if len(lk) != 4 * len(lookups):
        raise ValueError("grand_products: four columns per lookup")
    pl = (C.c_void_p * max(1, len(lk)))(*[ffi.dev_ptr(t).value for t in lk])
    total = len(perm_chunks) + len(lookups)
    dev_ = (vals + lk)[0].device if total else None
    zs = [torch.empty(32 << k, dtype=torch.uint8, device=dev_) for _ in range(total)]
    pz = (C.c_void_p * max(1, total))(*[z.data_ptr() for z in zs])
    ffi.check(ffi.lib().sg_grand_products_dev(pv, ps, cols, C.c_uint32(len(perm_chunks)), pl, C.c_uint32(len(lookups)),
                                              ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)), C.c_uint32(k), C.c_size_t(usable_rows), pz,
                                              ffi.current_stream_ptr()))
    return zs[:len(perm_chunks)], zs[len(perm_chunks):]


def quotient_permutation(values, zs, cols, sigmas, chunk_len: int, l0, l_last, l_active, beta, gamma, y, k: int,
                         ext_k: int, last_rotation_abs: int):
    """fold the permutation argument's constraints into the quotient numerator `values` (device
    tensors over the extended coset, in place): values = values * y + term, in halo2's order"""
    ns, m = len(zs), len(cols)
    if len(sigmas) != m:
        raise ValueError("quotient_permutation: one sigma per column")
    for t in [values, l0, l_last, l_active, *zs, *cols, *sigmas]:
        if t.numel() != 32 << ext_k:
            raise ValueError("quotient_permutation: every array has 2^ext_k rows")
    pz = (C.c_void_p * ns)(*[z.data_ptr() for z in zs])
    pc = (C.c_void_p * m)(*[c.data_ptr() for c in cols])
    ps = (C.c_void_p * m)(*[s.data_ptr() for s in sigmas])
    ffi.check(ffi.lib().sg_quotient_permutation_dev(
        ffi.dev_ptr(values), pz, C.c_uint32(ns), pc, ps, C.c_uint32(m), C.c_uint32(chunk_len), ffi.dev_ptr(l0),
        ffi.dev_ptr(l_last), ffi.dev_ptr(l_active), ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)), ffi.ptr(ffi.u8(y)),
        C.c_uint32(k), C.c_uint32(ext_k), C.c_uint32(last_rotation_abs), ffi.current_stream_ptr()))
    return values


def quotient_permutation_cosets(values, zs, cols, sigmas, chunk_len: int, l0, l_last, l_active, beta, gamma, y, k: int,
                                 ext_k: int, n_cosets: int, last_rotation_abs: int):
    """quotient_permutation over coset-major arrays (EvaluationDomain.coeff_to_cosets_batch): n_cosets blocks of 2^k rows"""
    ns, m = len(zs), len(cols)
    if len(sigmas) != m:
        raise ValueError("quotient_permutation_cosets: one sigma per column")
    for t in [values, l0, l_last, l_active, *zs, *cols, *sigmas]:
        if t.numel() != (32 * n_cosets) << k:
            raise ValueError("quotient_permutation_cosets: every array has n_cosets * 2^k rows")
    pz = (C.c_void_p * ns)(*[z.data_ptr() for z in zs])
    pc = (C.c_void_p * m)(*[c.data_ptr() for c in cols])
    ps = (C.c_void_p * m)(*[s.data_ptr() for s in sigmas])
    ffi.check(ffi.lib().sg_quotient_permutation_cosets_dev(
        ffi.dev_ptr(values), pz, C.c_uint32(ns), pc, ps, C.c_uint32(m), C.c_uint32(chunk_len), ffi.dev_ptr(l0),
        ffi.dev_ptr(l_last), ffi.dev_ptr(l_active), ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)), ffi.ptr(ffi.u8(y)),
        C.c_uint32(k), C.c_uint32(ext_k), C.c_uint32(n_cosets), C.c_uint32(last_rotation_abs), ffi.current_stream_ptr()))
    return values


def quotient_lookup_cosets(values, z, permuted_input, permuted_table, inp, table, l0, l_last, l_active, beta, gamma, y,
                           k: int, n_cosets: int):
    """quotient_lookup over coset-major arrays of n_cosets * 2^k rows"""
    for t in [values, z, permuted_input, permuted_table, inp, table, l0, l_last, l_active]:
        if t.numel() != (32 * n_cosets) << k:
            raise ValueError("quotient_lookup_cosets: every array has n_cosets * 2^k rows")
    ffi.check(ffi.lib().sg_quotient_lookup_cosets_dev(
        ffi.dev_ptr(values), ffi.dev_ptr(z), ffi.dev_ptr(permuted_input), ffi.dev_ptr(permuted_table),
        ffi.dev_ptr(inp), ffi.dev_ptr(table), ffi.dev_ptr(l0), ffi.dev_ptr(l_last), ffi.dev_ptr(l_active),
        ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)), ffi.ptr(ffi.u8(y)), C.c_uint32(k), C.c_uint32(n_cosets),
        ffi.current_stream_ptr()))
    return values


def quotient_lookup(values, z, permuted_input, permuted_table, inp, table, l0, l_last, l_active, beta, gamma, y,
                    k: int, ext_k: int):
    """fold one lookup argument's five constraints into the quotient numerator (in place)"""
    for t in [values, z, permuted_input, permuted_table, inp, table, l0, l_last, l_active]:
        if t.numel() != 32 << ext_k:
            raise ValueError("quotient_lookup: every array has 2^ext_k rows")
    ffi.check(ffi.lib().sg_quotient_lookup_dev(
        ffi.dev_ptr(values), ffi.dev_ptr(z), ffi.dev_ptr(permuted_input), ffi.dev_ptr(permuted_table),
        ffi.dev_ptr(inp), ffi.dev_ptr(table), ffi.dev_ptr(l0), ffi.dev_ptr(l_last), ffi.dev_ptr(l_active),
        ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)), ffi.ptr(ffi.u8(y)), C.c_uint32(k), C.c_uint32(ext_k),
        ffi.current_stream_ptr()))
    return values


class _VS(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32), ("rotation", C.c_uint32)]


class _Calc(C.Structure):
    _fields_ = [("op", C.c_uint32), ("a", _VS), ("b", _VS), ("parts_offset", C.c_uint32), ("parts_len", C.c_uint32)]


class _Graph(C.Structure):
    _fields_ = [("constants", C.c_void_p), ("n_constants", C.c_uint32), ("rotations", C.c_void_p),
                ("n_rotations", C.c_uint32), ("calculations", C.c_void_p), ("n_calculations", C.c_uint32),
                ("horner_parts", C.c_void_p), ("n_horner_parts", C.c_uint32)]


# halo2's ValueSource / Calculation variants (plonk/evaluation.rs), numbered as in include/summa_gpu.h
CONSTANT, INTERMEDIATE, FIXED, ADVICE, INSTANCE, CHALLENGE, BETA, GAMMA, THETA, Y, PREVIOUS_VALUE = range(11)
ADD, SUB, MUL, SQUARE, DOUBLE, NEGATE, HORNER, STORE = range(8)


class GraphEvaluator:
    """host mirror of halo2's GraphEvaluator: constants, rotations and calculations are appended the way
    upstream's add_constant / add_rotation / add_calculation do (deduplicated, returning a value source)"""

    def __init__(self):
        self.constants = []      # 32-byte Montgomery Fr each (bytes)
        self.rotations = []
        self.calculations = []   # (op, a, b[, parts]); a / b / parts: (kind, index, rotation_index)

    def add_constant(self, fr32) -> tuple:
        b = bytes(ffi.u8(fr32))
        if b not in self.constants:
            self.constants.append(b)
        return (CONSTANT, self.constants.index(b), 0)

    def add_rotation(self, rot: int) -> int:
        if rot not in self.rotations:
            self.rotations.append(rot)
        return self.rotations.index(rot)

    def query(self, kind: int, column: int, rot: int = 0) -> tuple:
        return (kind, column, self.add_rotation(rot))

    def add_calculation(self, op: int, a, b=None, parts=None) -> tuple:
        cal = (op, tuple(a), tuple(b) if b is not None else (0, 0, 0)) + ((list(parts),) if parts is not None else ())
        if cal in self.calculations:
            return (INTERMEDIATE, self.calculations.index(cal), 0)
        self.calculations.append(cal)
        return (INTERMEDIATE, len(self.calculations) - 1, 0)

    def as_dict(self):
        consts = np.frombuffer(b"".join(self.constants), dtype=np.uint8) if self.constants else np.zeros(0, np.uint8)
        return {"constants": consts, "rotations": list(self.rotations), "calculations": list(self.calculations)}

    def _struct(self):
        consts = np.frombuffer(b"".join(self.constants), dtype=np.uint8).copy() if self.constants else np.zeros(32, np.uint8)
        rots = np.asarray(self.rotations if self.rotations else [0], dtype=np.int32)
        parts = []
        calcs = (_Calc * max(1, len(self.calculations)))()
        for i, cal in enumerate(self.calculations):
            calcs[i].op = cal[0]
            calcs[i].a = _VS(*cal[1])
            calcs[i].b = _VS(*cal[2])
            if len(cal) > 3:
                calcs[i].parts_offset = len(parts)
                calcs[i].parts_len = len(cal[3])
                parts.extend(cal[3])
        parr = (_VS * max(1, len(parts)))(*[_VS(*p) for p in parts])
        g = _Graph(consts.ctypes.data, len(self.constants), rots.ctypes.data, len(self.rotations), C.addressof(calcs),
                   len(self.calculations), C.addressof(parr), len(parts))
        return g, (consts, rots, calcs, parr)


def quotient_gates(values, graph: GraphEvaluator, fixed, advice, instance, challenges, beta, gamma, theta, y, k: int,
                   ext_k: int):
    """run the custom-gate program over the extended coset: values[row] = program(row) (in place)"""
    for t in [values, *fixed, *advice, *instance]:
        if t.numel() != 32 << ext_k:
            raise ValueError("quotient_gates: every array has 2^ext_k rows")
    g, keep = graph._struct()
    arr = lambda ts: (C.c_void_p * max(1, len(ts)))(*[t.data_ptr() for t in ts])
    ch = np.ascontiguousarray(challenges, dtype=np.uint8) if len(challenges) else np.zeros(32, dtype=np.uint8)
    ffi.check(ffi.lib().sg_quotient_gates_dev(
        ffi.dev_ptr(values), C.byref(g), arr(fixed), C.c_uint32(len(fixed)), arr(advice), C.c_uint32(len(advice)),
        arr(instance), C.c_uint32(len(instance)), ffi.ptr(ch), C.c_uint32(len(challenges) // 32 if len(challenges) else 0),
        ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)), ffi.ptr(ffi.u8(theta)), ffi.ptr(ffi.u8(y)), C.c_uint32(k),
        C.c_uint32(ext_k), ffi.current_stream_ptr()))
    return values


def quotient_gates_cosets(values, graph: GraphEvaluator, fixed, advice, instance, challenges, beta, gamma, theta, y, k: int,
                          n_cosets: int):
    """quotient_gates over coset-major arrays of n_cosets * 2^k rows (a rotation stays inside its block)"""
    for t in [values, *fixed, *advice, *instance]:
        if t.numel() != (32 * n_cosets) << k:
            raise ValueError("quotient_gates_cosets: every array has n_cosets * 2^k rows")
    g, keep = graph._struct()
    arr = lambda ts: (C.c_void_p * max(1, len(ts)))(*[t.data_ptr() for t in ts])
    ch = np.ascontiguousarray(challenges, dtype=np.uint8) if len(challenges) else np.zeros(32, dtype=np.uint8)
    ffi.check(ffi.lib().sg_quotient_gates_cosets_dev(
        ffi.dev_ptr(values), C.byref(g), arr(fixed), C.c_uint32(len(fixed)), arr(advice), C.c_uint32(len(advice)),
        arr(instance), C.c_uint32(len(instance)), ffi.ptr(ch), C.c_uint32(len(challenges) // 32 if len(challenges) else 0),
        ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)), ffi.ptr(ffi.u8(theta)), ffi.ptr(ffi.u8(y)), C.c_uint32(k),
        C.c_uint32(n_cosets), ffi.current_stream_ptr()))
    return values


def quotient_numerator_cosets(values, gates: GraphEvaluator, lookup_input: GraphEvaluator, fixed, advice, instance, challenges, zs,
                              perm_cols, sigmas, chunk_len: int, l0, l_last, l_active, lookup_z, permuted_input, permuted_table,
                              table, beta, gamma, theta, y, k: int, ext_k: int, n_cosets: int, last_rotation_abs: int,
                              input_work=None):
    """halo2's evaluate_h over coset-major arrays in one call: values <- custom gates (previous value zero), permutation argument,
    lookup argument with `lookup_input` evaluated on the way (C ABI: sg_quotient_numerator_cosets_dev -- one fused pass for the
    reference circuit's programs, the separate kernels otherwise; the same words either way)"""
    ns, m = len(zs), len(perm_cols)
    if len(sigmas) != m:
        raise ValueError("quotient_numerator_cosets: one sigma per column")
    for t in [values, l0, l_last, l_active, lookup_z, permuted_input, permuted_table, table, *zs, *perm_cols, *sigmas, *fixed, *advice, *instance]:
        if t.numel() != (32 * n_cosets) << k:
            raise ValueError("quotient_numerator_cosets: every array has n_cosets * 2^k rows")
    g, keep = gates._struct()
    gi, keep_i = lookup_input._struct()
    arr = lambda ts: (C.c_void_p * max(1, len(ts)))(*[t.data_ptr() for t in ts])
    ch = np.ascontiguousarray(challenges, dtype=np.uint8) if len(challenges) else np.zeros(32, dtype=np.uint8)
    ffi.check(ffi.lib().sg_quotient_numerator_cosets_dev(
        ffi.dev_ptr(values), C.byref(g), C.byref(gi), arr(fixed), C.c_uint32(len(fixed)), arr(advice), C.c_uint32(len(advice)),
        arr(instance), C.c_uint32(len(instance)), ffi.ptr(ch), C.c_uint32(len(challenges) // 32 if len(challenges) else 0),
        arr(zs), C.c_uint32(ns), arr(perm_cols), arr(sigmas), C.c_uint32(m), C.c_uint32(chunk_len), ffi.dev_ptr(l0), ffi.dev_ptr(l_last),
        ffi.dev_ptr(l_active), ffi.dev_ptr(lookup_z), ffi.dev_ptr(permuted_input), ffi.dev_ptr(permuted_table), ffi.dev_ptr(table),
        ffi.dev_ptr(input_work) if input_work is not None else None, ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)), ffi.ptr(ffi.u8(theta)),
        ffi.ptr(ffi.u8(y)), C.c_uint32(k), C.c_uint32(ext_k), C.c_uint32(n_cosets), C.c_uint32(last_rotation_abs), ffi.current_stream_ptr()))
    return values


def fr_lincomb_sets(sets, n: int, outs=None):
    """several linear combinations of one length in ONE launch: sets = [(polys, coeffs (m x 32 B), low (<= 4 x 32 B or None)), ...];
    returns the output tensors (C ABI: sg_fr_lincomb_sets_dev)"""
    import torch
    polys = [p for ps, _, _ in sets for p in ps]
    for p in polys:
        if p.numel() != 32 * n:
            raise ValueError("fr_lincomb_sets: every polynomial has n coefficients")
    coeffs = np.concatenate([ffi.u8(c).reshape(-1) for _, c, _ in sets])
    sizes = (C.c_uint32 * len(sets))(*[len(ps) for ps, _, _ in sets])
    lows = np.zeros((len(sets), 4, 32), dtype=np.uint8)
    n_lows = (C.c_uint32 * len(sets))()
    for i, (_, _, low) in enumerate(sets):
        if low is not None and len(low):
            lw = ffi.u8(low).reshape(-1, 32)
            lows[i, :len(lw)] = lw
            n_lows[i] = len(lw)
    if outs is None:
        outs = [torch.empty(32 * n, dtype=torch.uint8, device=polys[0].device) for _ in sets]
    pp = (C.c_void_p * len(polys))(*[p.data_ptr() for p in polys])
    po = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
    ffi.check(ffi.lib().sg_fr_lincomb_sets_dev(pp, ffi.ptr(coeffs), sizes, C.c_uint32(len(sets)), C.c_size_t(n), ffi.ptr(lows.reshape(-1)),
                                               n_lows, po, ffi.current_stream_ptr()))
    return outs


def gates_program_info(graph: GraphEvaluator, n_fixed: int, n_advice: int, n_instance: int, n_challenges: int):
    """(instructions per row, simultaneously live values = LDS slots per row) of the interpreter's lowering; host only"""
    g, keep = graph._struct()
    n_ops, n_slots = C.c_uint32(0), C.c_uint32(0)
    ffi.check(ffi.lib().sg_gates_program_info(C.byref(g), C.c_uint32(n_fixed), C.c_uint32(n_advice), C.c_uint32(n_instance),
                                              C.c_uint32(n_challenges), C.byref(n_ops), C.byref(n_slots)))
    return n_ops.value, n_slots.value


def gates_program_words(graph: GraphEvaluator, n_fixed: int, n_advice: int, n_instance: int, n_challenges: int):
    """the interpreter's lowering of a program as words: [n_slots, result_kind, result_index, n_ops, (w0, dst, a, b) per
    instruction]; host only (tools/gen_gates_programs.py, tests)"""
    g, keep = graph._struct()
    need = C.c_uint32(0)
    args = (C.byref(g), C.c_uint32(n_fixed), C.c_uint32(n_advice), C.c_uint32(n_instance), C.c_uint32(n_challenges))
    ffi.check(ffi.lib().sg_gates_program_words(*args, None, C.c_uint32(0), C.byref(need)))
    buf = (C.c_uint32 * need.value)()
    ffi.check(ffi.lib().sg_gates_program_words(*args, buf, C.c_uint32(need.value), C.byref(need)))
    return list(buf)


def best_fft_batch(vectors, omega, log_n: int, divisor=None):
    """in-place best_fft (or ifft when `divisor` is given) of several device tensors of one size"""
    m = len(vectors)
    ptrs = (C.c_void_p * m)(*[v.data_ptr() for v in vectors])
    for v in vectors:
        if v.numel() != 32 << log_n:
            raise ValueError("best_fft: a.len() != 1 << log_n")
    dv = ffi.ptr(ffi.u8(divisor)) if divisor is not None else None
    ffi.check(ffi.lib().sg_ntt_fr_batch_dev(ptrs, C.c_size_t(m), ffi.ptr(ffi.u8(omega)), dv, C.c_uint32(log_n),
                                            ffi.current_stream_ptr()))
    return vectors
