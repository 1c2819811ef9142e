"""`create_proof` for the constraint system of the reference's MstInclusionCircuit (mst_inclusion.py), with every
data-parallel step on the GPU through the C ABI and the EVM / Keccak transcript of the generated verifier.

Host mirror of `halo2_proofs::plonk::create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<_>, ..>` as the
reference calls it [REF zk_prover/src/circuits/utils.rs:94-101, 171-178], restricted to what this constraint
system needs (one phase, one lookup, six permutation columns in two chunks, five quotient pieces).  The steps
and their order follow SURVEY.md §3.1 / Appendix C; the byte layout of the proof is the one the generated verifier
reads [REF contracts/src/InclusionVerifier.sol:274-367].  What stays on the host is what upstream also does serially
on the CPU: the transcript, the lookup's sort (`permute_expression_pair`), scalars of the multi-open; blinding
values come from a per-proof OS-random key expanded by ChaCha20 on the device (`sg_fr_random_dev`).

Inputs are a proving key made from Lagrange-basis fixed and permutation columns and an assignment of the three
advice columns (rows beyond n - 6 are overwritten with blinding values); key generation for the *reference's own*
circuit layout (its floor plan) is not restated -- tests/test_gpu_prover.py builds a satisfying assignment of the
same constraint system with its own fixed columns and checks the proof with the restated verifier
(oracle/summa_verifier.py), which also accepts the reference's shipped proof.
"""
from __future__ import annotations

import os
import threading

import numpy as np

from . import arithmetic as A
from . import mst_inclusion as M
from .domain import EvaluationDomain
from .merkle_sum_tree import keccak256
from .utils import ints_to_fr

R = M.R
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
DELTA = pow(7, 1 << 28, R)
ROT_LAST = -(M.BLINDING_FACTORS + 1)

# evaluations written to the proof, in order [REF InclusionVerifier.sol:500-1000, `calldataload` slots 0x03e4 ..]
EVAL_ORDER = ([("a", 0, 0), ("a", 1, 0), ("a", 0, 1), ("a", 1, 1), ("a", 2, 0), ("a", 1, -1), ("a", 0, -1),
               ("f", 2, 0), ("f", 3, 0), ("f", 0, 0), ("f", 1, 0)] + [("f", j, 0) for j in range(4, 11)] +
              [("random", 0, 0)] + [("sigma", j, 0) for j in range(6)] +
              [("z", 0, 0), ("z", 0, 1), ("z", 0, ROT_LAST), ("z", 1, 0), ("z", 1, 1),
               ("lz", 0, 0), ("lz", 0, 1), ("pin", 0, 0), ("pin", 0, -1), ("ptab", 0, 0)])
# rotation sets of the multi-open in nu order, polynomials of a set in increasing power of zeta [REF :1159-1340]
ROTATION_SETS = [
    ((-1, 0, 1), [("a", 0), ("a", 1)]),
    ((0,), [("a", 2), ("ptab", 0), ("f", 2), ("f", 3), ("f", 0), ("f", 1)] + [("f", j) for j in range(4, 11)] +
           [("sigma", j) for j in range(6)] + [("h", None), ("random", 0)]),
    ((ROT_LAST, 0, 1), [("z", 0)]),
    ((0, 1), [("z", 1), ("lz", 0)]),
    ((-1, 0), [("pin", 0)]),
]


def _inv(a):
    return pow(a, -1, R)


def _fr_bytes(v: int) -> np.ndarray:
    return ints_to_fr([v])


def _ints(t):
    """device Montgomery tensor -> list of Python ints"""
    raw = A.fr_from_montgomery(t).cpu().numpy().tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]


def _head(values, n: int):
    """device column of n rows: the given integers, then zeros"""
    import torch
    t = torch.zeros(32 * n, dtype=torch.uint8, device="cuda")
    if values:
        t[:32 * len(values)] = torch.from_numpy(ints_to_fr(values)).cuda()
    return t


def _canonical_rows(t) -> np.ndarray:
    """device Montgomery column -> (rows, 4) uint64 little-endian limbs of the canonical integers (host)"""
    return A.fr_from_montgomery(t).cpu().numpy().view(np.uint64).reshape(-1, 4)


def _sort_rows(limbs: np.ndarray) -> np.ndarray:
    """rows of 4 LE uint64 limbs in increasing integer order"""
    return limbs[np.lexsort((limbs[:, 0], limbs[:, 1], limbs[:, 2], limbs[:, 3]))]


def _point(c64: np.ndarray):
    """64-byte Montgomery affine -> (x, y) integers"""
    raw = bytes(c64)
    rinv = pow(1 << 256, -1, Q)
    return int.from_bytes(raw[:32], "little") * rinv % Q, int.from_bytes(raw[32:], "little") * rinv % Q


class EvmTranscriptWriter:
    """Keccak256 transcript + proof stream of halo2_solidity_verifier's `Keccak256Transcript` [REF :85-110], the
    flavour `gen_proof_solidity_calldata` proves with [REF zk_prover/src/circuits/utils.rs:170].  `vk.hash_into` is
    `common_scalar(vk digest)`: the buffer starts with it [REF InclusionVerifier.sol:217, 280-296]."""

    def __init__(self, vk_digest: int | None = None):
        self.buf = b""
        self.proof = bytearray()
        self._squeezed = False
        if vk_digest is not None:
            self.common_scalar(vk_digest)

    def common_scalar(self, v: int):
        self.buf += v.to_bytes(32, "big")
        self._squeezed = False

    def write_scalar(self, v: int):
        self.common_scalar(v)
        self.proof += v.to_bytes(32, "big")

    def common_point(self, p):
        if p is None:
            raise ValueError("cannot write points at infinity to the transcript")
        self.buf += p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")
        self._squeezed = False

    def write_point(self, p):
        self.common_point(p)
        self.proof += p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")

    def squeeze_challenge(self) -> int:
        """keccak(buffer) mod r, the hash becomes the buffer; a squeeze right after a squeeze hashes `hash || 0x01`"""
        h = keccak256(bytes(self.buf[:32]) + b"\x01") if self._squeezed else keccak256(bytes(self.buf))
        self.buf = h
        self._squeezed = True
        return int.from_bytes(h, "big") % R

    squeeze_challenge_again = squeeze_challenge

    def finalize(self) -> bytes:
        return bytes(self.proof)


def compress_g1(p) -> bytes:
    """halo2curves `G1Affine::to_bytes` (GroupEncoding): x as 32 bytes little-endian, bit 6 of the last byte = the
    parity of y, bit 7 = the point at infinity (bn256 Fq leaves two spare bits)  [UPSTREAM-RECALL: halo2curves 0.1.0
    derive/curve.rs `new_curve_impl!`; no fixture of the reference holds a compressed point]"""
    if p is None:
        return bytes(31) + b"\x80"
    enc = bytearray(p[0].to_bytes(32, "little"))
    enc[31] |= (p[1] & 1) << 6
    return bytes(enc)


class Blake2bWrite:
    """`Blake2bWrite<_, G1Affine, Challenge255<_>>`, the transcript `full_prover` proves with
    [REF zk_prover/src/circuits/utils.rs:93-101] (halo2_proofs transcript.rs): Blake2b-512
    personalised "Halo2-Transcript"; a point is absorbed as prefix 1 || x || y (32-byte little-endian canonical
    reprs) and written to the proof compressed (32 B), a scalar as prefix 2 || repr; a challenge absorbs prefix 0,
    finalises a CLONE of the state to 64 bytes and reduces them as a 512-bit little-endian integer mod r
    (`from_uniform_bytes`).  The hash itself is the standard library's (as upstream takes it from blake2b_simd);
    the oracle carries an independent implementation pinned on RFC 7693 (tests/test_transcript_cpu.py)."""

    def __init__(self, vk_digest: int | None = None):
        import hashlib
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.proof = bytearray()
        if vk_digest is not None:
            self.common_scalar(vk_digest)

    def common_scalar(self, v: int):
        self.state.update(b"\x02" + v.to_bytes(32, "little"))

    def write_scalar(self, v: int):
        self.common_scalar(v)
        self.proof += v.to_bytes(32, "little")

    def common_point(self, p):
        if p is None:
            raise ValueError("cannot write points at infinity to the transcript")
        self.state.update(b"\x01" + p[0].to_bytes(32, "little") + p[1].to_bytes(32, "little"))

    def write_point(self, p):
        self.common_point(p)
        self.proof += compress_g1(p)

    def squeeze_challenge(self) -> int:
        self.state.update(b"\x00")
        return int.from_bytes(self.state.copy().digest(), "little") % R

    squeeze_challenge_again = squeeze_challenge

    def finalize(self) -> bytes:
        return bytes(self.proof)


def verifying_key_digest(k: int, n_currencies: int, fixed_comms, permutation_comms) -> int:
    """The scalar `vk.hash_into` feeds the transcript: halo2's `VerifyingKey::transcript_repr`, i.e. Blake2b-512
    (personalised "Halo2-Verify-Key") of  len(s) || s  with s = the `{:?}` text of the pinned verification key, reduced
    from 64 bytes mod r.  `vk_repr` rebuilds that text -- the constraint system as `MstInclusionConfig::configure`
    leaves it, expression trees included -- so the value equals a Rust-built key's: for MstInclusionCircuit<4,2,8>,
    k = 11 and the reference's SRS it is the `vk_digest` constant of contracts/src/InclusionVerifier.sol:217
    (tests/test_api_cpu.py pins that)."""
    from . import vk_repr
    return vk_repr.transcript_repr(k, n_currencies, fixed_comms, permutation_comms)


class ProvingKey:
    """fixed and permutation columns in the three bases halo2's pk keeps (Lagrange, coefficients, extended coset),
    the Lagrange selector polynomials l0 / l_last / l_active, and the verifying key's commitments"""

    def __init__(self, params, k: int, fixed_lagrange, sigma_lagrange, n_currencies: int = 2, quotient_domain: str | None = None):
        import torch
        # where the quotient is evaluated: "cosets" = the first degree - 1 cosets of the extended domain (all that h
        # needs; include/summa_gpu.h: sg_coeff_to_cosets_batch_dev), "extended" = halo2's whole extended domain
        self.quotient_domain = quotient_domain or os.environ.get("SUMMA_QUOTIENT_DOMAIN", "cosets")
        if self.quotient_domain not in ("cosets", "extended"):
            raise ValueError("quotient_domain: cosets or extended")
        self.k, self.n = k, 1 << k
        self._lock = threading.Lock()             # guards the caches made on first use (native key, witness program)
        self.n_currencies = n_currencies          # one sum gate per currency in the gate program
        self.dom = EvaluationDomain(M.DEGREE, k)
        n, u = self.n, self.n - (M.BLINDING_FACTORS + 1)
        self.usable_rows = u
        assert len(fixed_lagrange) == M.NUM_FIXED and len(sigma_lagrange) == len(M.PERMUTATION_COLUMNS)
        self.fixed_lagrange, self.sigma_lagrange = list(fixed_lagrange), list(sigma_lagrange)
        one = torch.from_numpy(ints_to_fr([1])).cuda()
        l_last = torch.zeros(32 * n, dtype=torch.uint8, device="cuda")
        l_last[32 * u:32 * (u + 1)] = one
        sel = [_head([1], n), l_last, torch.cat([one.repeat(u), torch.zeros(32 * (n - u), dtype=torch.uint8, device="cuda")])]
        cols = self.fixed_lagrange + self.sigma_lagrange + sel
        coeff = A.best_fft_batch([c.clone() for c in cols], self.dom.get_omega_inv(), k, divisor=self.dom.ifft_divisor())
        ext = (self.dom.coeff_to_cosets_batch if self.quotient_domain == "cosets" else self.dom.coeff_to_extended_batch)(coeff)
        nf, ns = M.NUM_FIXED, len(sigma_lagrange)
        self.fixed_coeff, self.sigma_coeff = coeff[:nf], coeff[nf:nf + ns]
        self.fixed_ext, self.sigma_ext = ext[:nf], ext[nf:nf + ns]
        self.l0_ext, self.l_last_ext, self.l_active_ext = ext[nf + ns:]
        comms = params.commit_batch(self.fixed_lagrange + self.sigma_lagrange, lagrange=True)
        self.fixed_comms = [_point(c) for c in comms[:nf]]
        self.permutation_comms = [_point(c) for c in comms[nf:]]
        self.vk_digest = verifying_key_digest(k, n_currencies, self.fixed_comms, self.permutation_comms)
        torch.cuda.synchronize()


    def free(self) -> None:
        """give back the key's forms inside the library's compiled prover (sp_key_create on first use: coefficient and
        coset columns of every fixed / permutation column, on the device); the key object itself stays usable and
        makes them again when a proof asks.  Called when the object dies: a long-lived process that proves for one
        snapshot after another must not keep every past key's columns (tests/test_gpu_batch.py, the long-lived-process test)."""
        made = getattr(self, "_native", None)
        if made is None:
            return
        self._native = None
        try:
            import ctypes as C
            from . import ffi
            ffi.prover_lib().sp_key_destroy(C.c_uint64(made[0]))
        except Exception:   # noqa: BLE001 -- interpreter shutdown, library gone: nothing left to give back to
            pass

    def __del__(self):
        self.free()


_KEY_LOCK = threading.Lock()      # for key objects that carry no lock of their own


def native_key(pk: ProvingKey, params) -> int:
    """the proving key inside the library's compiled prover (sp_key_create), made on first use and kept on `pk`.
    Several proofs may start on a fresh key at once (batch.prove_batch's worker threads): the key is made once, under
    the key's lock, and the others wait for it."""
    made = getattr(pk, "_native", None)
    if made is not None and made[1] == (params.handle(), pk.vk_digest):
        return made[0]
    with getattr(pk, "_lock", _KEY_LOCK):
        return _native_key_locked(pk, params)


def _native_key_locked(pk: ProvingKey, params) -> int:
    import ctypes as C
    from . import ffi
    made = getattr(pk, "_native", None)
    if made is not None and made[1] == (params.handle(), pk.vk_digest):
        return made[0]
    if made is not None:
        ffi.check_prover(ffi.prover_lib().sp_key_destroy(C.c_uint64(made[0])))
    gates, keep1 = M.gate_graph(pk.n_currencies)._struct()
    look, keep2 = M.lookup_input_graph()._struct()
    fixed = (C.c_void_p * len(pk.fixed_lagrange))(*[c.data_ptr() for c in pk.fixed_lagrange])
    sigma = (C.c_void_p * len(pk.sigma_lagrange))(*[c.data_ptr() for c in pk.sigma_lagrange])
    digest = np.frombuffer(pk.vk_digest.to_bytes(32, "big"), dtype=np.uint8).copy()
    key = C.c_uint64(0)
    groups = M.gate_challenge_exponents(pk.n_currencies)
    exps = (C.c_uint32 * max(1, sum(len(g) for g in groups)))(*[e for g in groups for e in g])
    counts = (C.c_uint32 * max(1, len(groups)))(*[len(g) for g in groups])
    ffi.check_prover(ffi.prover_lib().sp_key_create(C.c_uint32(pk.k), C.c_uint64(params.handle()), fixed, sigma, ffi.ptr(digest),
                                                    C.byref(gates), C.byref(look), exps, counts, C.c_uint32(len(groups)),
                                                    ffi.current_stream_ptr(), C.byref(key)))
    pk._native = (key.value, (params.handle(), pk.vk_digest))
    return key.value


def create_proof_native(params, pk: ProvingKey, advice, instances, flavour: str = "evm", sanity_checks: bool = True,
                        in_place: bool = False, combine: bool = False) -> bytes:
    """`create_proof` by the library's compiled host driver (include/summa_prover.hpp behind the C ABI of
    include/summa_prover.h): same steps, same transcripts, same proofs as create_proof below, without the interpreter
    between the kernels -- and with the GIL released for the whole proof, so several can be in flight from Python
    threads.  advice: 3 device columns (copied unless in_place); instances: integers; flavour "evm" | "blake2b"."""
    import ctypes as C
    from . import ffi
    if len(advice) != M.NUM_ADVICE or any(a.numel() != 32 * pk.n for a in advice):
        raise ValueError(f"create_proof: {M.NUM_ADVICE} advice columns of 2^k rows expected")
    if len(instances) > pk.usable_rows or any(not 0 <= v < R for v in instances):
        raise ValueError("create_proof: instances are field elements on usable rows")
    if flavour not in ("evm", "blake2b"):
        raise ValueError("flavour: evm or blake2b")
    key = native_key(pk, params)
    cols = list(advice) if in_place else [a.clone() for a in advice]
    ptrs = (C.c_void_p * 3)(*[c.data_ptr() for c in cols])
    inst = ints_to_fr(list(instances)) if instances else np.zeros(0, dtype=np.uint8)
    out = np.zeros(2144, dtype=np.uint8)
    size = C.c_size_t(0)
    # combine: this proof is one of several in flight on other threads (batch.prove_batch) -- its commitment jobs may be
    # fused with theirs by the library's commit combiner (include/summa_gpu.h: sg_commit_combine_begin)
    if combine:
        ffi.check(ffi.lib().sg_commit_combine_begin())
    try:
        rc = ffi.prover_lib().sp_create_proof(C.c_uint64(key), ptrs, ffi.ptr(inst) if len(instances) else None, C.c_uint32(len(instances)),
                                              C.c_int(0 if flavour == "evm" else 1), C.c_int(1 if sanity_checks else 0),
                                              ffi.current_stream_ptr(), ffi.ptr(out), C.c_size_t(out.size), C.byref(size))
    finally:
        if combine:
            ffi.check(ffi.lib().sg_commit_combine_end())
    if rc == -6:      # SG_ERR_WITNESS
        raise ValueError(ffi.prover_lib().sp_last_error().decode())
    ffi.check_prover(rc)
    return out[:size.value].tobytes()


def permute_expression_pair(inp: np.ndarray, table: np.ndarray):
    """halo2 `lookup::prover::permute_expression_pair` on the usable rows, as (rows, 4) uint64 limb arrays of
    canonical integers: A' = the input sorted, S' = the table rearranged so that every row has A'[i] == S'[i] or
    A'[i] == A'[i-1] (the leftover table values fill the repeated rows).  Host work, as upstream."""
    a = _sort_rows(inp)
    first = np.ones(len(a), dtype=bool)
    first[1:] = (a[1:] != a[:-1]).any(axis=1)
    # multiset difference  table - {distinct input values}; one-limb tables (range checks) take the 1-D path
    small = not table[:, 1:].any()
    if small and a[:, 1:].any():
        raise ValueError("lookup input value not in the table")
    s = np.empty_like(a)
    s[first] = a[first]
    if small:
        t_sorted, distinct = np.sort(table[:, 0]), a[first, 0]
        at = np.searchsorted(t_sorted, distinct)
        if (at >= len(t_sorted)).any() or (t_sorted[np.minimum(at, len(t_sorted) - 1)] != distinct).any():
            raise ValueError("lookup input value not in the table")
        keep = np.ones(len(t_sorted), dtype=bool)
        keep[at] = False                                   # one occurrence per distinct input value
        s[~first] = 0
        s[~first, 0] = t_sorted[keep]
    else:
        both, inverse = np.unique(np.concatenate([table, a[first]]), axis=0, return_inverse=True)
        inverse = inverse.reshape(-1)
        have = np.bincount(inverse[:len(table)], minlength=len(both))
        need = np.bincount(inverse[len(table):], minlength=len(both))
        if (have < need).any():
            raise ValueError("lookup input value not in the table")
        s[~first] = np.repeat(both, have - need, axis=0)
    return a, s


def create_proof(params, pk: ProvingKey, advice, instances, seed: bytes | None = None, timings=None, transcript=None,
                 sanity_checks: bool = True) -> bytes:
    """advice: 3 device tensors (Lagrange, 2^k rows, Montgomery Fr); instances: list of ints -> proof bytes.
    `seed`: 32-byte ChaCha20 key for the blinding factors and the random polynomial (tests); default: the OS
    entropy source.  `transcript`: a fresh EvmTranscriptWriter (default; `gen_proof_solidity_calldata`'s flavour) or
    Blake2bWrite (`full_prover`'s) -- upstream's `create_proof` takes it as an argument the same way.
    `sanity_checks` (upstream's cargo feature of that name): refuse an assignment whose permutation or lookup grand
    product does not close; without it such a witness yields a proof the verifier rejects, as upstream's default
    build does (the reference's test of a wrong public input relies on that, circuits/tests.rs:125-152).  A lookup
    input outside the table always raises (upstream: `permute_expression_pair` fails the prover)."""
    import torch
    # blinding values: a 32-byte key from the OS entropy source per proof, expanded by ChaCha20 on the device
    # (sg_fr_random_dev); every draw takes its own stream id
    import os
    key = seed if seed is not None else os.urandom(32)
    draws = [0]

    def rand(count):
        draws[0] += 1
        return A.fr_random(key, draws[0], count)
    import time
    clock = [time.perf_counter()]

    def lap(name):   # per-phase wall clock (device drained), only when asked for
        if timings is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (now - clock[0]) * 1e3
            clock[0] = now
    k, n, u, dom = pk.k, pk.n, pk.usable_rows, pk.dom
    if len(advice) != M.NUM_ADVICE or any(a.numel() != 32 * n for a in advice):
        raise ValueError(f"create_proof: {M.NUM_ADVICE} advice columns of 2^k rows expected")
    if len(instances) > u or any(not 0 <= v < R for v in instances):
        raise ValueError("create_proof: instances are field elements on usable rows")
    if seed is not None and len(seed) != 32:
        raise ValueError("create_proof: the seed is a 32-byte key")
    ext_k = dom.extended_k
    ne = 1 << ext_k
    omega = pow(ROOT_OF_UNITY, 1 << (28 - k), R)
    none = np.zeros(0, dtype=np.uint8)
    tr = transcript if transcript is not None else EvmTranscriptWriter()
    tr.common_scalar(pk.vk_digest)                    # vk.hash_into(transcript)
    for v in instances:
        tr.common_scalar(v)
    polys = {}      # key -> coefficient-form device polynomial (n coefficients)

    def to_coeff(cols):
        return A.best_fft_batch([c.clone() for c in cols], dom.get_omega_inv(), k, divisor=dom.ifft_divisor())

    # -- 1: advice columns: blind the last rows, commit
    noncanonical = A.count_noncanonical(advice) if sanity_checks else None     # read where the host waits next
    advice = [a.clone() for a in advice]
    for a in advice:
        a[32 * u:] = rand(n - u)
    instance_col = _head(list(instances), n)
    # -- 2, computed ahead of its place in the transcript: the lookup's permuted pair.  One input and one table expression:
    # the theta-compression is the expression itself, nothing here waits for theta, so the two permuted columns share ONE
    # fused commitment job with the advice columns; their points are written where upstream writes them (after theta).
    inp_d = torch.zeros(32 * n, dtype=torch.uint8, device="cuda")
    A.quotient_gates(inp_d, M.lookup_input_graph(), pk.fixed_lagrange, advice, [instance_col], none,
                     _fr_bytes(0), _fr_bytes(0), _fr_bytes(0), _fr_bytes(0), k, k)   # the expression row by row (stride 1)
    on_device = A.lookup_permute_small(inp_d, pk.fixed_lagrange[4], u)   # range tables: no host round trip
    if on_device is not None:
        pin_d, ptab_d = (torch.cat([col, rand(n - u)]) for col in on_device)             # blinding rows random
    else:
        pin_rows, ptab_rows = permute_expression_pair(_canonical_rows(inp_d)[:u], _canonical_rows(pk.fixed_lagrange[4])[:u])
        pin_d, ptab_d = (torch.cat([A.fr_to_montgomery(torch.from_numpy(rows.view(np.uint8).reshape(-1)).cuda()), rand(n - u)])
                         for rows in (pin_rows, ptab_rows))
    points = [_point(c) for c in params.commit_batch_mixed(advice + [pin_d, ptab_d], [1, 1, 1, 2, 2])]   # sorted columns: flag 2
    for p in points[:3]:
        tr.write_point(p)
    if noncanonical is not None and int(noncanonical.item()):
        raise ValueError("create_proof: advice words >= r (not canonical Montgomery field elements)")
    theta = tr.squeeze_challenge()  # one expression per side: theta only separates the phases
    adv_coeff = to_coeff(advice + [instance_col])
    for j in range(3):
        polys[("a", j)] = adv_coeff[j]
    on_cosets = pk.quotient_domain == "cosets"
    to_ext = dom.coeff_to_cosets_batch if on_cosets else dom.coeff_to_extended_batch
    ext1 = to_ext(adv_coeff)
    adv_ext, inst_ext = ext1[:3], ext1[3]

    lap("1_advice")
    for p in points[3:]:
        tr.write_point(p)
    beta = tr.squeeze_challenge()
    gamma = tr.squeeze_challenge_again()
    b_beta, b_gamma, b_theta = _fr_bytes(beta), _fr_bytes(gamma), _fr_bytes(theta)

    lap("2_lookup")
    # -- 3: grand products (device scans), blinding rows, commitments; then the random polynomial
    col_lag = {(A.ADVICE, j): advice[j] for j in range(3)}
    col_lag.update({(A.FIXED, j): pk.fixed_lagrange[j] for j in range(M.NUM_FIXED)})
    col_lag[(A.INSTANCE, 0)] = instance_col
    zs, last, delta_start = [], None, 1
    for c0 in range(0, len(M.PERMUTATION_COLUMNS), M.PERMUTATION_CHUNK):
        chunk = M.PERMUTATION_COLUMNS[c0:c0 + M.PERMUTATION_CHUNK]
        z = A.permutation_product([col_lag[c] for c in chunk], pk.sigma_lagrange[c0:c0 + len(chunk)], b_beta, b_gamma,
                                  _fr_bytes(delta_start), k, z0=None if last is None else _fr_bytes(last))
        last = _ints(z[32 * u:32 * (u + 1)])[0]
        z[32 * (u + 1):] = rand(n - u - 1)
        zs.append(z)
        delta_start = delta_start * pow(DELTA, len(chunk), R) % R
    if sanity_checks and last != 1:
        raise ValueError("permutation argument not satisfied by the assignment")
    lz = A.lookup_product(inp_d, pk.fixed_lagrange[4], pin_d, ptab_d, b_beta, b_gamma)
    if sanity_checks and _ints(lz[32 * u:32 * (u + 1)])[0] != 1:
        raise ValueError("lookup argument not satisfied by the assignment")
    lz[32 * (u + 1):] = rand(n - u - 1)
    polys[("random", 0)] = rand(n)
    # one fused job; the grand products are constant over the unused rows (every ratio is 1 there): difference form
    for c in params.commit_batch_mixed(zs + [lz, polys[("random", 0)]], [2, 2, 2, 0]):
        tr.write_point(_point(c))
    y = tr.squeeze_challenge()
    b_y = _fr_bytes(y)
    co3 = to_coeff([pin_d, ptab_d] + zs + [lz])
    polys[("pin", 0)], polys[("ptab", 0)], polys[("z", 0)], polys[("z", 1)], polys[("lz", 0)] = co3
    pin_ext, ptab_ext, z0_ext, z1_ext, lz_ext = to_ext(co3)

    lap("3_grand_products")
    # -- 4: quotient: evaluate_h over the extended coset, / (X^n - 1), back to coefficients, five pieces
    col_ext = {(A.ADVICE, j): adv_ext[j] for j in range(3)}
    col_ext.update({(A.FIXED, j): pk.fixed_ext[j] for j in range(M.NUM_FIXED)})
    col_ext[(A.INSTANCE, 0)] = inst_ext
    perm_cols = [col_ext[c] for c in M.PERMUTATION_COLUMNS]
    if on_cosets:
        # deg h < 5 n: its values on 5 cosets of the 2^k domain determine it; the kernels take the coset-major arrays whole
        # (a rotation is an index shift of 1 inside a block), and the pieces come straight out of sg_cosets_to_pieces_dev
        d = dom.quotient_poly_degree
        values = torch.zeros(32 * n * d, dtype=torch.uint8, device="cuda")
        A.quotient_gates_cosets(values, M.gate_graph(pk.n_currencies), pk.fixed_ext, adv_ext, [inst_ext], M.gate_challenges(y, pk.n_currencies),
                                b_beta, b_gamma, b_theta, b_y, k, d)
        A.quotient_permutation_cosets(values, [z0_ext, z1_ext], perm_cols, pk.sigma_ext, M.PERMUTATION_CHUNK, pk.l0_ext, pk.l_last_ext,
                                      pk.l_active_ext, b_beta, b_gamma, b_y, k, ext_k, d, M.BLINDING_FACTORS + 1)
        input_c = torch.empty(32 * n * d, dtype=torch.uint8, device="cuda")
        A.quotient_gates_cosets(input_c, M.lookup_input_graph(), pk.fixed_ext, adv_ext, [inst_ext], none, b_beta, b_gamma, b_theta, b_y, k, d)
        A.quotient_lookup_cosets(values, lz_ext, pin_ext, ptab_ext, input_c, pk.fixed_ext[4], pk.l0_ext, pk.l_last_ext, pk.l_active_ext,
                                 b_beta, b_gamma, b_y, k, d)
        pieces = dom.cosets_to_pieces(values)
    else:
        values = torch.zeros(32 * ne, dtype=torch.uint8, device="cuda")
        A.quotient_gates(values, M.gate_graph(pk.n_currencies), pk.fixed_ext, adv_ext, [inst_ext], M.gate_challenges(y, pk.n_currencies), b_beta, b_gamma, b_theta,
                         b_y, k, ext_k)
        A.quotient_permutation(values, [z0_ext, z1_ext], perm_cols, pk.sigma_ext, M.PERMUTATION_CHUNK, pk.l0_ext, pk.l_last_ext,
                               pk.l_active_ext, b_beta, b_gamma, b_y, k, ext_k, M.BLINDING_FACTORS + 1)
        input_ext = torch.zeros(32 * ne, dtype=torch.uint8, device="cuda")
        A.quotient_gates(input_ext, M.lookup_input_graph(), pk.fixed_ext, adv_ext, [inst_ext], none, b_beta,
                         b_gamma, b_theta, b_y, k, ext_k)
        A.quotient_lookup(values, lz_ext, pin_ext, ptab_ext, input_ext, pk.fixed_ext[4], pk.l0_ext, pk.l_last_ext, pk.l_active_ext,
                          b_beta, b_gamma, b_y, k, ext_k)
        dom.divide_by_vanishing_poly(values)
        h = dom.extended_to_coeff(values)
        pieces = [h[32 * n * i:32 * n * (i + 1)].clone() for i in range(M.DEGREE - 1)]
    for c in params.commit_batch(pieces):
        tr.write_point(_point(c))
    x = tr.squeeze_challenge()
    x_n = pow(x, n, R)

    lap("4_quotient")
    # -- 5: evaluations
    for j in range(M.NUM_FIXED):
        polys[("f", j)] = pk.fixed_coeff[j]
    for j in range(len(pk.sigma_coeff)):
        polys[("sigma", j)] = pk.sigma_coeff[j]
    point = lambda rot: x * pow(omega, rot % n, R) % R
    ev = A.eval_polynomial_batch([polys[(kind, c)] for kind, c, _ in EVAL_ORDER] + pieces,     # the quotient pieces ride along
                                 np.concatenate([_fr_bytes(point(rot)) for _, _, rot in EVAL_ORDER] + [_fr_bytes(x)] * len(pieces)))
    rinv = pow(1 << 256, -1, R)
    evals = {key: int.from_bytes(bytes(e), "little") * rinv % R for key, e in zip(EVAL_ORDER, ev)}
    for key in EVAL_ORDER:
        tr.write_scalar(evals[key])
    polys[("h", None)] = A.lincomb(pieces, np.concatenate([_fr_bytes(pow(x_n, i, R)) for i in range(len(pieces))]))
    h_eval = 0                                                    # h(x) = sum_i x^(n i) h_i(x)
    for e in reversed(ev[len(EVAL_ORDER):]):
        h_eval = (h_eval * x_n + int.from_bytes(bytes(e), "little") * rinv) % R

    def eval_of(key, rot):
        return h_eval if key == ("h", None) else evals[(key[0], key[1], rot)]

    lap("5_evaluations")
    # -- 6: SHPLONK (BDFG21) multi-open
    zeta = tr.squeeze_challenge()
    nu = tr.squeeze_challenge_again()
    zero_row = torch.zeros(32, dtype=torch.uint8, device="cuda")
    qs, rs, fs, div_in, div_pts, weights = [], [], [], [], [], []
    for si, (rots, keys) in enumerate(ROTATION_SETS):
        q = A.lincomb([polys[key] for key in keys], np.concatenate([_fr_bytes(pow(zeta, j, R)) for j in range(len(keys))]))
        pts = [point(r) for r in rots]
        vals = [sum(pow(zeta, j, R) * eval_of(key, r) for j, key in enumerate(keys)) % R for r in rots]
        # r(X): the polynomial of degree < |S| through (pts, vals) (Lagrange interpolation on integers)
        r_coeff = [0] * len(pts)
        denoms = []
        for i, (pi, vi) in enumerate(zip(pts, vals)):
            basis, denom = [1], 1
            for j, pj in enumerate(pts):
                if j != i:
                    basis = [((basis[t - 1] if t else 0) - pj * (basis[t] if t < len(basis) else 0)) % R for t in range(len(basis) + 1)]
                    denom = denom * (pi - pj) % R
            denoms.append(_inv(denom))
            scale = vi * denoms[-1] % R
            for t, b in enumerate(basis):
                r_coeff[t] = (r_coeff[t] + scale * b) % R
        r_poly = _head(r_coeff, n)
        f = A.lincomb([q, r_poly], np.concatenate([_fr_bytes(1), _fr_bytes(R - 1)]))
        # (q - r) / Z_S: q - r vanishes on the whole set and 1 / prod_j (X - p_j) = sum_j c_j / (X - p_j) with the Lagrange
        # denominators c_j = 1 / prod_{t != j} (p_j - p_t): every division of every set is an independent exact Kate division
        for pj, cj in zip(pts, denoms):
            div_in.append(f)
            div_pts.append(_fr_bytes(pj))
            weights.append(_fr_bytes(pow(nu, si, R) * cj % R))
        qs.append(q)
        rs.append(r_coeff)
        fs.append(f)
    quotients = A.kate_division_batch(div_in, np.concatenate(div_pts))            # all eleven in one batch
    f_all = A.lincomb(quotients, np.concatenate(weights))                          # sum_i nu^i (q_i - r_i) / Z_{S_i}
    w = _point(params.commit(f_all))
    tr.write_point(w)
    mu = tr.squeeze_challenge()
    all_rots = sorted({r for rots, _ in ROTATION_SETS for r in rots})
    mu_minus = {r: (mu - point(r)) % R for r in all_rots}
    diffs = []
    for rots, _ in ROTATION_SETS:
        d = 1
        for r in all_rots:
            if r not in rots:
                d = d * mu_minus[r] % R
        diffs.append(d)
    d0_inv = _inv(diffs[0])
    z_s0 = 1
    for r in ROTATION_SETS[0][0]:
        z_s0 = z_s0 * mu_minus[r] % R
    coeffs, const = [], 0
    for i, (d, r_coeff) in enumerate(zip(diffs, rs)):
        scale = pow(nu, i, R) * d % R * d0_inv % R
        coeffs.append(scale)
        r_at_mu = sum(c * pow(mu, t, R) for t, c in enumerate(r_coeff)) % R
        const = (const + scale * r_at_mu) % R
    one_poly = _head([1], n)
    l_poly = A.lincomb(qs + [f_all, one_poly],
                       np.concatenate([_fr_bytes(c) for c in coeffs] + [_fr_bytes((-z_s0) % R), _fr_bytes((-const) % R)]))
    quo, rem = A.kate_division(l_poly, _fr_bytes(mu), with_remainder=True)
    if bytes(rem) != bytes(32):
        raise ValueError("multi-open linearisation does not vanish at mu")
    tr.write_point(_point(params.commit(torch.cat([quo, zero_row]))))
    lap("6_multiopen")
    return tr.finalize()


def export_bundle(path: str, params, pk: ProvingKey, advice, instances) -> None:
    """everything the C++ driver (include/summa_prover.hpp, tools/create_proof_main.cpp) needs for one proof: SRS,
    the proving key's Lagrange columns, the two GraphEvaluator programs, the assignment and the public inputs"""
    import struct

    def graph_bytes(g: A.GraphEvaluator) -> bytes:
        out = struct.pack("<I", len(g.constants)) + b"".join(g.constants)
        out += struct.pack("<I", len(g.rotations)) + struct.pack(f"<{len(g.rotations)}i", *g.rotations)
        parts, calcs = [], b""
        for cal in g.calculations:
            off, ln = 0, 0
            if len(cal) > 3:
                off, ln = len(parts), len(cal[3])
                parts.extend(cal[3])
            calcs += struct.pack("<9I", cal[0], *cal[1], *cal[2], off, ln)
        out += struct.pack("<I", len(g.calculations)) + calcs
        out += struct.pack("<I", len(parts)) + b"".join(struct.pack("<3I", *p) for p in parts)
        return out
    host = lambda t: t.cpu().numpy().tobytes()
    with open(path, "wb") as f:
        f.write(b"SGPB2\0\0\0" + struct.pack("<II", pk.k, len(instances)))
        f.write(params.g.tobytes() + params.g_lagrange.tobytes())
        for col in pk.fixed_lagrange + pk.sigma_lagrange + list(advice):
            f.write(host(col))
        f.write(ints_to_fr(list(instances)).tobytes())
        f.write(pk.vk_digest.to_bytes(32, "big"))
        f.write(graph_bytes(M.gate_graph(pk.n_currencies)) + graph_bytes(M.lookup_input_graph()))
        groups = M.gate_challenge_exponents(pk.n_currencies)      # the gate program's challenges: sums of powers of y
        f.write(struct.pack("<I", len(groups)) + b"".join(struct.pack("<I%dI" % len(g), len(g), *g) for g in groups))
