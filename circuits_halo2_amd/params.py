"""Mirror of `halo2_proofs::poly::kzg::commitment::ParamsKZG<Bn256>` for the methods the
reference calls (zk_prover/src/circuits/utils.rs:55 read, :58-66 k()/downsize, :70 setup) and
the two commitment methods every MSM of the prover goes through (commit, commit_lagrange).

SRS container layout (halo2 `SerdeFormat::RawBytes`):
    k:u32 LE || g[2^k] || g_lagrange[2^k] || g2 || s_g2      (G1 64 B, G2 128 B, Montgomery)
The bases are uploaded once and stay resident in HBM (handle in the C ABI's SRS cache).
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import ffi
from .arithmetic import _is_torch_cuda, g1_fixed_base_mul


class ParamsKZG:
    def __init__(self, k: int, g: np.ndarray, g_lagrange: np.ndarray, g2: bytes = b"", s_g2: bytes = b""):
        self.k = k
        self.n = 1 << k
        self._g = ffi.u8(g)
        self._g_lagrange = ffi.u8(g_lagrange)
        if self._g.size != 64 * self.n or self._g_lagrange.size != 64 * self.n:
            raise ValueError("ParamsKZG: basis length != 2^k")
        self.g2, self.s_g2 = g2, s_g2
        self._handle = None

    @classmethod
    def from_device(cls, k: int, d_g, d_g_lagrange, g2: bytes = b"", s_g2: bytes = b"") -> "ParamsKZG":
        """params whose bases arrive in device memory (the receive side of the setup broadcast, batch.py): they go into the
        library's SRS cache device to device (sg_srs_upload_dev) and visit the host only if `g` / `g_lagrange` / `write`
        are asked for"""
        if not (_is_torch_cuda(d_g) and _is_torch_cuda(d_g_lagrange)) or d_g.numel() != 64 << k or d_g_lagrange.numel() != 64 << k:
            raise ValueError("ParamsKZG.from_device: two device buffers of 2^k x 64 bytes expected")
        self = cls.__new__(cls)
        self.k, self.n = k, 1 << k
        self._g = self._g_lagrange = None
        self.g2, self.s_g2 = g2, s_g2
        h = C.c_uint64(0)
        ffi.check(ffi.lib().sg_srs_upload_dev(C.c_uint32(k), ffi.dev_ptr(d_g), ffi.dev_ptr(d_g_lagrange), ffi.current_stream_ptr(),
                                              C.byref(h)))
        self._handle = h.value
        return self

    def device_bases(self):
        """copies of the resident g / g_lagrange as two device tensors (the send side of the setup broadcast)"""
        import torch
        d_g = torch.empty(64 * self.n, dtype=torch.uint8, device="cuda")
        d_gl = torch.empty(64 * self.n, dtype=torch.uint8, device="cuda")
        ffi.check(ffi.lib().sg_srs_copy_dev(C.c_uint64(self.handle()), ffi.dev_ptr(d_g), ffi.dev_ptr(d_gl), ffi.current_stream_ptr()))
        return d_g, d_gl

    def _fetch_host(self):
        if self._g is None:
            d_g, d_gl = self.device_bases()
            self._g, self._g_lagrange = d_g.cpu().numpy(), d_gl.cpu().numpy()

    @property
    def g(self) -> np.ndarray:
        self._fetch_host()
        return self._g

    @property
    def g_lagrange(self) -> np.ndarray:
        self._fetch_host()
        return self._g_lagrange

    # --- construction -------------------------------------------------------------------
    @classmethod
    def read(cls, reader) -> "ParamsKZG":
        """ParamsKZG::read (RawBytes).  `reader`: bytes or a binary file object."""
        raw = reader if isinstance(reader, (bytes, bytearray)) else reader.read()
        if len(raw) < 4:
            raise ValueError("Failed to read params")
        (k,) = struct.unpack_from("<I", raw, 0)
        n = 1 << k
        if k > 28 or len(raw) != 4 + 2 * n * 64 + 256:
            raise ValueError("Failed to read params")
        buf = np.frombuffer(raw, dtype=np.uint8)
        return cls(k, buf[4:4 + 64 * n].copy(), buf[4 + 64 * n:4 + 128 * n].copy(),
                   bytes(raw[4 + 128 * n:4 + 128 * n + 128]), bytes(raw[4 + 128 * n + 128:]))

    @classmethod
    def setup(cls, k: int, tau) -> "ParamsKZG":
        """ParamsKZG::setup(k, rng): the "unsafe" setup of generate_setup_artifacts(k, None, ..)
        (utils.rs:68-71).  `tau`: the secret scalar as 32 B Montgomery Fr (upstream draws it
        from OsRng).  Scalars tau^i / L_i(tau) and the 2 * 2^k fixed-base products run on the GPU."""
        t = ffi.u8(tau)
        if t.size != 32:
            raise ValueError("tau must be one 32-byte Fr")
        g = np.zeros(64 << k, dtype=np.uint8)
        gl = np.zeros(64 << k, dtype=np.uint8)
        ffi.check(ffi.lib().sg_kzg_setup(C.c_uint32(k), ffi.ptr(t), ffi.ptr(g), ffi.ptr(gl)))
        # verifier side: g2 = G2 generator, s_g2 = tau * g2 (two scalar multiplications over Fq2, host code)
        one = np.zeros(32, dtype=np.uint8)
        one[:] = np.frombuffer(bytes.fromhex("fbffff4f1c3496ac29cd609f9576fc362e4679786fa36e662fdf079ac1770a0e"), dtype=np.uint8)
        g2, s_g2 = np.zeros(128, dtype=np.uint8), np.zeros(128, dtype=np.uint8)
        ffi.check(ffi.lib().sg_g2_generator_mul(ffi.ptr(one), ffi.ptr(g2)))
        ffi.check(ffi.lib().sg_g2_generator_mul(ffi.ptr(t), ffi.ptr(s_g2)))
        return cls(k, g, gl, g2.tobytes(), s_g2.tobytes())

    def write(self) -> bytes:
        """ParamsKZG::write (RawBytes): k || g || g_lagrange || g2 || s_g2 -- the container `read` parses"""
        if len(self.g2) != 128 or len(self.s_g2) != 128:
            raise ValueError("params without g2 / s_g2 cannot be serialised")
        return struct.pack("<I", self.k) + self.g.tobytes() + self.g_lagrange.tobytes() + bytes(self.g2) + bytes(self.s_g2)

    @classmethod
    def setup_from_tau_powers(cls, k: int, tau_powers_mont: np.ndarray, lagrange_evals_mont: np.ndarray):
        """ParamsKZG::setup's group part: g[i] = tau^i * G, g_lagrange[i] = L_i(tau) * G, both
        as fixed-base products on the GPU (the scalar side -- powers of tau and L_i(tau) -- is
        supplied by the caller)."""
        return cls(k, g1_fixed_base_mul(tau_powers_mont), g1_fixed_base_mul(lagrange_evals_mont))

    def downsize(self, k: int) -> None:
        """ParamsKZG::downsize(k) (utils.rs:62-65): keep g[0..2^k) and recompute g_lagrange for
        the smaller domain with an inverse FFT over G1 (sg_g1_to_lagrange)."""
        if k > self.k:
            raise ValueError("k is too large for the given params")  # utils.rs:58-60
        if k == self.k:
            return
        g_all = self.g          # (fetched from the device first when the params came from there)
        self.free()
        n = 1 << k
        g = np.ascontiguousarray(g_all[:64 * n])
        gl = np.zeros(64 * n, dtype=np.uint8)
        ffi.check(ffi.lib().sg_g1_to_lagrange(ffi.ptr(g), C.c_uint32(k), ffi.ptr(gl)))
        self.k, self.n, self._g, self._g_lagrange = k, n, g, gl

    # --- device cache -------------------------------------------------------------------
    def handle(self) -> int:
        if self._handle is None:
            h = C.c_uint64(0)
            ffi.check(ffi.lib().sg_srs_upload(C.c_uint32(self.k), ffi.ptr(self.g), ffi.ptr(self.g_lagrange),
                                              C.byref(h)))
            self._handle = h.value
        return self._handle

    def check(self) -> None:
        """what `ParamsKZG::read` checks with SerdeFormat::RawBytes and this mirror's `read` leaves to the first use of
        the device: every point of g / g_lagrange is on the curve (halo2's `from_raw_bytes`); raises "Failed to read
        params" otherwise.  `read(..)` itself stays byte-level (`RawBytesUnchecked`) so that it works without a GPU."""
        bad = C.c_uint64(0)
        ffi.check(ffi.lib().sg_srs_check(C.c_uint64(self.handle()), C.byref(bad)))
        if bad.value:
            raise ValueError(f"Failed to read params: {bad.value} points are not on the curve")

    def precompute(self, basis: int | None = None, window_bits: int = 0) -> None:
        """build the fixed-base window table(s) of the resident SRS (sg_srs_precompute): later
        commit / commit_lagrange / commit_batch calls take the fixed-base path (same results).
        basis: 0 = g, 1 = g_lagrange, 2 = the prefix sums of g_lagrange (difference-form commitments of piecewise-constant
        Lagrange columns: `commit_batch(..., diff=True)`, flag 2 of `commit_batch_mixed`), None = all three."""
        for b in ([0, 1, 2] if basis is None else [basis]):
            ffi.check(ffi.lib().sg_srs_precompute(C.c_uint64(self.handle()), C.c_int(b), C.c_uint32(window_bits)))

    def free(self):
        if self._handle is not None:
            self._fetch_host()    # the host copy is what a later handle() uploads again
            ffi.check(ffi.lib().sg_srs_free(C.c_uint64(self._handle)))
            self._handle = None

    # --- commitments --------------------------------------------------------------------
    def _commit(self, basis: int, poly):
        L = ffi.lib()
        out = np.zeros(64, dtype=np.uint8)
        if _is_torch_cuda(poly):
            n = poly.numel() // 32
            if n > self.n:
                raise ValueError("polynomial longer than the SRS")
            ffi.check(L.sg_commit_dev(C.c_uint64(self.handle()), C.c_int(basis), ffi.dev_ptr(poly), C.c_size_t(n),
                                      ffi.current_stream_ptr(), ffi.ptr(out)))
            return out
        s = ffi.u8(poly)
        if s.size % 32 or s.size // 32 > self.n:
            raise ValueError("polynomial longer than the SRS")
        ffi.check(L.sg_commit(C.c_uint64(self.handle()), C.c_int(basis), ffi.ptr(s), C.c_size_t(s.size // 32),
                              ffi.ptr(out)))
        return out

    def commit_batch(self, polys, lagrange: bool = False, diff: bool = False) -> np.ndarray:
        """commitments to several equal-length device polynomials as fused jobs -> (len, 64) uint8.
        diff (Lagrange columns only): a hint that the columns are piecewise constant -- the library commits to the
        differences against the prefix-summed basis when that table exists (sg_commit, basis 2); same commitments."""
        m = len(polys)
        out = np.zeros((m, 64), dtype=np.uint8)
        if m == 0:
            return out
        n = polys[0].numel() // 32
        for p in polys:
            if not _is_torch_cuda(p) or p.numel() != 32 * n:
                raise ValueError("commit_batch: equal-length device tensors expected")
        if n > self.n:
            raise ValueError("polynomial longer than the SRS")
        ptrs = (C.c_void_p * m)(*[p.data_ptr() for p in polys])
        ffi.check(ffi.lib().sg_commit_batch_dev(C.c_uint64(self.handle()), C.c_int((2 if diff else 1) if lagrange else 0), ptrs,
                                                C.c_size_t(m), C.c_size_t(n), ffi.current_stream_ptr(), ffi.ptr(out)))
        return out

    def commit_batch_mixed(self, polys, lagrange_flags) -> np.ndarray:
        """like commit_batch with one basis per polynomial (False / 0 = coefficients, True / 1 = Lagrange form, 2 = Lagrange
        form with the piecewise-constant hint; any of them | 16 = SG_BASIS_SPARSE, the "witness-like column" scheduling hint
        of include/summa_gpu.h): one fused job for a phase that commits to both kinds"""
        m = len(polys)
        out = np.zeros((m, 64), dtype=np.uint8)
        if m == 0:
            return out
        if len(lagrange_flags) != m:
            raise ValueError("commit_batch_mixed: one flag per polynomial")
        n = polys[0].numel() // 32
        for p in polys:
            if not _is_torch_cuda(p) or p.numel() != 32 * n:
                raise ValueError("commit_batch_mixed: equal-length device tensors expected")
        if n > self.n:
            raise ValueError("polynomial longer than the SRS")
        ptrs = (C.c_void_p * m)(*[p.data_ptr() for p in polys])
        flags = (C.c_int * m)(*[int(f) for f in lagrange_flags])
        if any((f & ~16) not in (0, 1, 2) for f in flags):
            raise ValueError("commit_batch_mixed: flags are 0, 1 or 2, optionally | 16 (SG_BASIS_SPARSE)")
        ffi.check(ffi.lib().sg_commit_batch_mixed_dev(C.c_uint64(self.handle()), flags, ptrs, C.c_size_t(m), C.c_size_t(n),
                                                      ffi.current_stream_ptr(), ffi.ptr(out)))
        return out

    def commit(self, poly):
        """commit to a polynomial in coefficient form: best_multiexp(poly, g)"""
        return self._commit(0, poly)

    def commit_lagrange(self, poly):
        """commit to a polynomial in Lagrange form: best_multiexp(poly, g_lagrange)"""
        return self._commit(1, poly)
