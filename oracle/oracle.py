"""ctypes loader for oracle/liboracle.so (C restatement; test infrastructure only).

All buffers are numpy uint8 arrays in the halo2curves in-memory layout: Fr = 32 B
(4 x u64 LE limbs, Montgomery), G1Affine = 64 B (x||y Montgomery Fq, identity = zeros).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    if os.environ.get("SUMMA_ORACLE_LIB"):        # another build of the same source (tests/test_sanitizers_cpu.py: liboracle_asan.so)
        return os.environ["SUMMA_ORACLE_LIB"]
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "bn254_oracle.c")
    if force or not os.path.exists(so) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_best_multiexp.restype = C.c_int
        _LIB.orc_g1_is_on_curve.restype = C.c_int
    return _LIB


def _p(a: np.ndarray):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _buf(nbytes: int) -> np.ndarray:
    return np.zeros(nbytes, dtype=np.uint8)


def ncpu() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        return os.cpu_count() or 1


def best_multiexp(scalars: np.ndarray, bases: np.ndarray, threads: int = 1) -> np.ndarray:
    n = scalars.size // 32
    assert bases.size == 64 * n
    out = _buf(64)
    lib().orc_best_multiexp(_p(scalars), _p(bases), C.c_size_t(n), C.c_int(threads), _p(out))
    return out


def best_fft(a: np.ndarray, omega: np.ndarray, log_n: int, threads: int = 1) -> np.ndarray:
    out = np.ascontiguousarray(a).copy()
    assert out.size == 32 << log_n
    lib().orc_best_fft(_p(out), _p(omega), C.c_uint32(log_n), C.c_int(threads))
    return out


def lagrange_to_coeff(a: np.ndarray, k: int, threads: int = 1) -> np.ndarray:
    out = np.ascontiguousarray(a).copy()
    assert out.size == 32 << k
    lib().orc_lagrange_to_coeff(_p(out), C.c_uint32(k), C.c_int(threads))
    return out


def ifft(a, omega_inv, divisor, log_n, threads=1):
    out = np.ascontiguousarray(a).copy()
    lib().orc_ifft(_p(out), _p(omega_inv), _p(divisor), C.c_uint32(log_n), C.c_int(threads))
    return out


def coeff_to_extended(coeffs: np.ndarray, k: int, ext_k: int, threads: int = 1) -> np.ndarray:
    assert coeffs.size == 32 << k
    out = _buf(32 << ext_k)
    lib().orc_coeff_to_extended(_p(coeffs), C.c_uint32(k), C.c_uint32(ext_k), _p(out), C.c_int(threads))
    return out


def extended_to_coeff(ext: np.ndarray, k: int, ext_k: int, threads: int = 1) -> np.ndarray:
    out = np.ascontiguousarray(ext).copy()
    assert out.size == 32 << ext_k
    lib().orc_extended_to_coeff(_p(out), C.c_uint32(k), C.c_uint32(ext_k), C.c_int(threads))
    return out


def divide_by_vanishing_poly(ext: np.ndarray, k: int, ext_k: int) -> np.ndarray:
    out = np.ascontiguousarray(ext).copy()
    lib().orc_divide_by_vanishing_poly(_p(out), C.c_uint32(k), C.c_uint32(ext_k))
    return out


def _const(fn, *args) -> np.ndarray:
    out = _buf(32)
    fn(*args, _p(out))
    return out


def omega(k: int) -> np.ndarray:
    return _const(lib().orc_omega, C.c_uint32(k))


def omega_inv(k: int) -> np.ndarray:
    return _const(lib().orc_omega_inv, C.c_uint32(k))


def n_inv(k: int) -> np.ndarray:
    return _const(lib().orc_n_inv, C.c_uint32(k))


def zeta() -> np.ndarray:
    return _const(lib().orc_zeta)


def _bin(fn, a, b, n_out=32):
    out = _buf(n_out)
    fn(_p(a), _p(b), _p(out))
    return out


def fr_mul(a, b):
    return _bin(lib().orc_fr_mul, a, b)


def fq_mul(a, b):
    return _bin(lib().orc_fq_mul, a, b)


def fr_add(a, b):
    return _bin(lib().orc_fr_add, a, b)


def fr_sub(a, b):
    return _bin(lib().orc_fr_sub, a, b)


def fr_inv(a):
    out = _buf(32)
    lib().orc_fr_inv(_p(a), _p(out))
    return out


def fr_to_mont(canon: np.ndarray) -> np.ndarray:
    out = _buf(canon.size)
    lib().orc_fr_to_mont_n(_p(canon), C.c_size_t(canon.size // 32), _p(out))
    return out


def fr_from_mont(m: np.ndarray) -> np.ndarray:
    out = _buf(m.size)
    lib().orc_fr_from_mont_n(_p(m), C.c_size_t(m.size // 32), _p(out))
    return out


def fr_dot(a, b) -> np.ndarray:
    out = _buf(32)
    lib().orc_fr_dot(_p(a), _p(b), C.c_size_t(a.size // 32), _p(out))
    return out


def fr_powers(tau: np.ndarray, n: int) -> np.ndarray:
    out = _buf(32 * n)
    lib().orc_fr_powers(_p(tau), C.c_size_t(n), _p(out))
    return out


def fr_eval_poly(coeffs, x) -> np.ndarray:
    out = _buf(32)
    lib().orc_fr_eval_poly(_p(coeffs), C.c_size_t(coeffs.size // 32), _p(x), _p(out))
    return out


def fr_batch_invert(a) -> np.ndarray:
    out = np.ascontiguousarray(a).copy()
    lib().orc_fr_batch_invert(_p(out), C.c_size_t(out.size // 32))
    return out


def fr_prefix_product(a) -> np.ndarray:
    n = a.size // 32
    out = _buf(32 * (n + 1))
    lib().orc_fr_prefix_product(_p(np.ascontiguousarray(a)), C.c_size_t(n), _p(out))
    return out


def fr_mul_n(a, b) -> np.ndarray:
    out = _buf(a.size)
    lib().orc_fr_mul_n(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), C.c_size_t(a.size // 32), _p(out))
    return out


def fr_kate_division(a, b):
    """-> (q: n - 1 coefficients, remainder a(b))"""
    a = np.ascontiguousarray(a)
    n = a.size // 32
    q, rem = _buf(32 * max(1, n - 1)), _buf(32)
    lib().orc_fr_kate_division(_p(a), C.c_size_t(n), _p(b), _p(q), _p(rem))
    return q[:32 * max(0, n - 1)], rem


def fr_lincomb(polys, coeffs) -> np.ndarray:
    ps = [np.ascontiguousarray(p) for p in polys]
    ptrs = (C.c_void_p * len(ps))(*[p.ctypes.data for p in ps])
    out = _buf(ps[0].size)
    lib().orc_fr_lincomb(ptrs, _p(np.ascontiguousarray(coeffs)), C.c_uint32(len(ps)), C.c_size_t(ps[0].size // 32), _p(out))
    return out


def permutation_product(values, sigmas, beta, gamma, delta_start, k, z0=None) -> np.ndarray:
    m = len(values)
    vs = [np.ascontiguousarray(v) for v in values]
    ss = [np.ascontiguousarray(s) for s in sigmas]
    pv = (C.c_void_p * m)(*[v.ctypes.data for v in vs])
    ps = (C.c_void_p * m)(*[s.ctypes.data for s in ss])
    out = _buf(32 << k)
    lib().orc_permutation_product(pv, ps, C.c_uint32(m), _p(beta), _p(gamma), _p(delta_start), C.c_uint32(k),
                                  _p(z0) if z0 is not None else None, _p(out))
    return out


def lookup_product(a, s, ap, sp, beta, gamma) -> np.ndarray:
    n = a.size // 32
    out = _buf(32 * n)
    lib().orc_lookup_product(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(s)), _p(np.ascontiguousarray(ap)),
                             _p(np.ascontiguousarray(sp)), _p(beta), _p(gamma), C.c_size_t(n), _p(out))
    return out


def set_quotient_threads(threads: int) -> None:
    """row loops of the three quotient blocks below on `threads` pthreads (upstream parallelises them with rayon)"""
    lib().orc_set_quotient_threads(C.c_int(threads))


def quotient_permutation(values, zs, cols, sigmas, chunk_len, l0, l_last, l_active, beta, gamma, y, k, ext_k,
                         last_rotation_abs) -> np.ndarray:
    """returns the updated numerator (values is not modified)"""
    out = np.ascontiguousarray(values).copy()
    keep = [np.ascontiguousarray(t) for t in [*zs, *cols, *sigmas]]
    ns, m = len(zs), len(cols)
    pz = (C.c_void_p * ns)(*[t.ctypes.data for t in keep[:ns]])
    pc = (C.c_void_p * m)(*[t.ctypes.data for t in keep[ns:ns + m]])
    ps = (C.c_void_p * m)(*[t.ctypes.data for t in keep[ns + m:]])
    lib().orc_quotient_permutation(_p(out), pz, C.c_uint32(ns), pc, ps, C.c_uint32(m), C.c_uint32(chunk_len),
                                   _p(np.ascontiguousarray(l0)), _p(np.ascontiguousarray(l_last)),
                                   _p(np.ascontiguousarray(l_active)), _p(beta), _p(gamma), _p(y), C.c_uint32(k),
                                   C.c_uint32(ext_k), C.c_uint32(last_rotation_abs))
    return out


def quotient_lookup(values, z, ap, sp, a, s, l0, l_last, l_active, beta, gamma, y, k, ext_k) -> np.ndarray:
    out = np.ascontiguousarray(values).copy()
    c = np.ascontiguousarray
    lib().orc_quotient_lookup(_p(out), _p(c(z)), _p(c(ap)), _p(c(sp)), _p(c(a)), _p(c(s)), _p(c(l0)), _p(c(l_last)),
                              _p(c(l_active)), _p(beta), _p(gamma), _p(y), C.c_uint32(k), C.c_uint32(ext_k))
    return out


class _VS(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32), ("rotation", C.c_uint32)]


class _Calc(C.Structure):
    _fields_ = [("op", C.c_uint32), ("a", _VS), ("b", _VS), ("parts_offset", C.c_uint32), ("parts_len", C.c_uint32)]


class _Graph(C.Structure):
    _fields_ = [("constants", C.c_void_p), ("n_constants", C.c_uint32), ("rotations", C.c_void_p),
                ("n_rotations", C.c_uint32), ("calculations", C.c_void_p), ("n_calculations", C.c_uint32),
                ("horner_parts", C.c_void_p), ("n_horner_parts", C.c_uint32)]


def _graph_struct(graph):
    """graph: {"constants": bytes-like n x 32, "rotations": [int], "calculations": [(op, a, b[, parts])]}
    with a / b / parts entries = (kind, index, rotation_index); returns (struct, keepalive)"""
    consts = np.ascontiguousarray(graph["constants"], dtype=np.uint8)
    rots = np.asarray(graph["rotations"], dtype=np.int32)
    parts = []
    calcs = (_Calc * max(1, len(graph["calculations"])))()
    for i, cal in enumerate(graph["calculations"]):
        op, a, b = cal[0], cal[1], cal[2] if len(cal) > 2 and cal[2] is not None else (0, 0, 0)
        calcs[i].op = op
        calcs[i].a = _VS(*a)
        calcs[i].b = _VS(*b)
        if len(cal) > 3:
            calcs[i].parts_offset = len(parts)
            calcs[i].parts_len = len(cal[3])
            parts.extend(cal[3])
    parr = (_VS * max(1, len(parts)))(*[_VS(*p) for p in parts])
    g = _Graph(consts.ctypes.data, consts.size // 32, rots.ctypes.data, rots.size, C.addressof(calcs),
               len(graph["calculations"]), C.addressof(parr), len(parts))
    return g, (consts, rots, calcs, parr)


def quotient_gates(values, graph, fixed, advice, instance, challenges, beta, gamma, theta, y, k, ext_k) -> np.ndarray:
    out = np.ascontiguousarray(values).copy()
    g, keep = _graph_struct(graph)
    cols = [[np.ascontiguousarray(c) for c in group] for group in (fixed, advice, instance)]
    ptrs = [(C.c_void_p * max(1, len(group)))(*[c.ctypes.data for c in group]) for group in cols]
    ch = np.ascontiguousarray(challenges, dtype=np.uint8) if len(challenges) else np.zeros(32, dtype=np.uint8)
    lib().orc_quotient_gates(_p(out), C.byref(g), ptrs[0], ptrs[1], ptrs[2], _p(ch), _p(beta), _p(gamma), _p(theta), _p(y),
                             C.c_uint32(k), C.c_uint32(ext_k))
    return out


_PSD_READY = False


def _poseidon_ready():
    global _PSD_READY
    if not _PSD_READY:
        from . import poseidon_params, pyref
        rcs, mds, _ = poseidon_params.generate()
        rc = np.frombuffer(pyref.frs_to_bytes([x for r in rcs for x in r]), dtype=np.uint8).copy()
        md = np.frombuffer(pyref.frs_to_bytes([x for r in mds for x in r]), dtype=np.uint8).copy()
        lib().orc_poseidon_set_params(_p(rc), _p(md))
        _PSD_READY = True


def poseidon_hash(inputs: np.ndarray) -> np.ndarray:
    _poseidon_ready()
    out = _buf(32)
    lib().orc_poseidon_hash(_p(np.ascontiguousarray(inputs)), C.c_size_t(inputs.size // 32), _p(out))
    return out


def mst_leaves(users: np.ndarray, balances: np.ndarray, nc: int) -> np.ndarray:
    _poseidon_ready()
    n = users.size // 32
    out = _buf(32 * n)
    lib().orc_mst_leaves(_p(np.ascontiguousarray(users)), _p(np.ascontiguousarray(balances)), C.c_size_t(n), C.c_uint32(nc), _p(out))
    return out


def mst_level(child_hash: np.ndarray, child_bal: np.ndarray, nc: int):
    _poseidon_ready()
    m = child_hash.size // 64
    h, b = _buf(32 * m), _buf(32 * m * nc)
    lib().orc_mst_level(_p(np.ascontiguousarray(child_hash)), _p(np.ascontiguousarray(child_bal)), C.c_size_t(m),
                        C.c_uint32(nc), _p(h), _p(b))
    return h, b


def g1_is_on_curve(p) -> bool:
    return bool(lib().orc_g1_is_on_curve(_p(np.ascontiguousarray(p))))


def g1_mul(p, scalar) -> np.ndarray:
    return _bin(lib().orc_g1_mul, np.ascontiguousarray(p), np.ascontiguousarray(scalar), 64)


def g1_add(p, q) -> np.ndarray:
    return _bin(lib().orc_g1_add, np.ascontiguousarray(p), np.ascontiguousarray(q), 64)


def g1_generator() -> np.ndarray:
    out = _buf(64)
    lib().orc_g1_generator(_p(out))
    return out


def fixed_base_mul(scalars: np.ndarray, threads: int = 1) -> np.ndarray:
    n = scalars.size // 32
    out = _buf(64 * n)
    lib().orc_fixed_base_mul(_p(scalars), C.c_size_t(n), C.c_int(threads), _p(out))
    return out


def random_fr(seed: int, n: int) -> np.ndarray:
    out = _buf(32 * n)
    lib().orc_random_fr(C.c_uint64(seed), C.c_size_t(n), _p(out))
    return out
