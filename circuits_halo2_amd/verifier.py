"""`verify_proof` for the constraint system of MstInclusionCircuit -- what `full_verifier` and
`create_proof_checked` run on a proof [REF zk_prover/src/circuits/utils.rs:110-131, 181-193]: halo2's
`verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK, _, _, SingleStrategy>`.

Host side: the transcript replay, the Lagrange / instance evaluations, the constraint polynomials at x (the product's
own expression list, mst_inclusion.gates / lookup_expressions), the SHPLONK scalars.  Group side: every commitment
enters the final check linearly, so the whole left-hand side is ONE multi-scalar multiplication of ~37 points on the
device (C ABI `sg_msm_g1`) followed by the two-pairing check `sg_pairing_check`:

    e( sum_i c_i C_i  -  r G  -  Z_{S_0}(mu) W  +  mu W',  [1]_2 ) * e( -W', [s]_2 ) == 1

with the same scaling by 1 / Z_{T \\ S_0}(mu) as the reference's generated verifier
[REF contracts/src/InclusionVerifier.sol:1025-1402].  Both transcript flavours: "evm" (Keccak, 2144-byte proofs) and
"blake2b" (Challenge255, 1632-byte proofs with compressed points).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import ffi
from . import mst_inclusion as M
from . import prover as P
from .arithmetic import ADVICE, FIXED, INSTANCE, best_multiexp

R, Q = P.R, P.Q
_QR = pow(1 << 256, 1, Q)


def _inv(a: int) -> int:
    return pow(a, -1, R)


def _g1_bytes(p) -> bytes:
    """(x, y) integers -> 64-byte Montgomery affine (identity: None -> zeros)"""
    if p is None:
        return bytes(64)
    return (p[0] * _QR % Q).to_bytes(32, "little") + (p[1] * _QR % Q).to_bytes(32, "little")


def _fr_bytes(v: int) -> bytes:
    return ((v % R) << 256).__mod__(R).to_bytes(32, "little")


def _on_curve(p) -> bool:
    return p[0] < Q and p[1] < Q and (p[1] * p[1] - p[0] * p[0] * p[0] - 3) % Q == 0


def decompress_g1(enc: bytes):
    """inverse of prover.compress_g1; ValueError for anything that is not a finite curve point"""
    last = enc[31]
    if last & 0x80:
        raise ValueError("point at infinity")
    x = int.from_bytes(enc[:31] + bytes([last & 0x3F]), "little")
    if x >= Q:
        raise ValueError("x coordinate not reduced")
    y2 = (x * x * x + 3) % Q
    y = pow(y2, (Q + 1) // 4, Q)
    if y * y % Q != y2:
        raise ValueError("not on the curve")
    return (x, Q - y if (y & 1) != ((last >> 6) & 1) else y)


class _Reader:
    """transcript + proof reader of either flavour (`Keccak256Transcript::read_*` / `Blake2bRead::read_*`)"""

    def __init__(self, proof: bytes, flavour: str):
        if flavour not in ("evm", "blake2b"):
            raise ValueError("flavour: evm or blake2b")
        self.evm = flavour == "evm"
        self.tr = P.EvmTranscriptWriter() if self.evm else P.Blake2bWrite()
        self.proof, self.pos = proof, 0

    def _take(self, size: int) -> bytes:
        if self.pos + size > len(self.proof):
            raise ValueError("proof too short")
        out = self.proof[self.pos:self.pos + size]
        self.pos += size
        return out

    def read_point(self):
        if self.evm:
            raw = self._take(64)
            p = (int.from_bytes(raw[:32], "big"), int.from_bytes(raw[32:], "big"))
            if not _on_curve(p):
                raise ValueError("commitment not on the curve")
        else:
            p = decompress_g1(self._take(32))
        self.tr.common_point(p)
        return p

    def read_scalar(self) -> int:
        v = int.from_bytes(self._take(32), "big" if self.evm else "little")
        if v >= R:
            raise ValueError("scalar not reduced")
        self.tr.common_scalar(v)
        return v


def _lagrange(x: int, k: int, instances):
    """l_0, l_last, l_blind (the five blinding rows) and the instance column's value at x"""
    n = 1 << k
    omega = pow(P.ROOT_OF_UNITY, 1 << (28 - k), R)
    x_n = pow(x, n, R)
    common = (x_n - 1) * _inv(n) % R
    li = lambda i: common * pow(omega, i % n, R) % R * _inv((x - pow(omega, i % n, R)) % R) % R
    return {"x_n": x_n, "l_0": li(0), "l_last": li(P.ROT_LAST), "l_blind": sum(li(i) for i in range(P.ROT_LAST + 1, 0)) % R,
            "instance": sum(li(i) * v for i, v in enumerate(instances)) % R}


def _expected_h_eval(evals, ch, lag, n_currencies: int) -> int:
    """all constraints folded with y in the constraint system's order (gates, permutation, lookup), / (x^n - 1)"""
    beta, gamma, y, x = ch["beta"], ch["gamma"], ch["y"], ch["x"]
    l_0, l_last = lag["l_0"], lag["l_last"]
    active = (1 - l_last - lag["l_blind"]) % R
    kinds = {ADVICE: "a", FIXED: "f"}

    def query(kind, column, rotation):
        return lag["instance"] if kind == INSTANCE else evals[(kinds[kind], column, rotation)]
    terms = [g.evaluate(query) for g in M.gates(n_currencies)]
    cols = M.PERMUTATION_COLUMNS
    chunks = [cols[i:i + M.PERMUTATION_CHUNK] for i in range(0, len(cols), M.PERMUTATION_CHUNK)]
    z = lambda j, rot=0: evals[("z", j, rot)]
    last = len(chunks) - 1
    terms.append(l_0 * (1 - z(0)) % R)
    terms.append(l_last * (z(last) * z(last) - z(last)) % R)
    for j in range(1, len(chunks)):
        terms.append(l_0 * (z(j) - z(j - 1, P.ROT_LAST)) % R)
    shift, col = beta * x % R, 0
    for j, chunk in enumerate(chunks):
        left, right = z(j, 1), z(j)
        for kind, c in chunk:
            v = query(kind, c, 0)
            left = left * (v + beta * evals[("sigma", col, 0)] + gamma) % R
            right = right * (v + shift + gamma) % R
            shift = shift * P.DELTA % R
            col += 1
        terms.append((left - right) * active % R)
    inp_e, tab_e = M.lookup_expressions()
    inp, tab = inp_e.evaluate(query), tab_e.evaluate(query)
    lz, lz_next = evals[("lz", 0, 0)], evals[("lz", 0, 1)]
    pin, pin_prev, ptab = evals[("pin", 0, 0)], evals[("pin", 0, -1)], evals[("ptab", 0, 0)]
    terms.append(l_0 * (1 - lz) % R)
    terms.append(l_last * (lz * lz - lz) % R)
    terms.append(active * (lz_next * (pin + beta) % R * (ptab + gamma) - lz * (inp + beta) % R * (tab + gamma)) % R)
    terms.append(l_0 * (pin - ptab) % R)
    terms.append(active * (pin - ptab) % R * (pin - pin_prev) % R)
    acc = 0
    for t in terms:
        acc = (acc * y + t) % R
    return acc * _inv((lag["x_n"] - 1) % R) % R


DRIVER = os.environ.get("SUMMA_VERIFIER_DRIVER", "native")   # "native": sp_verify_proof, the library's compiled verifier; "python": the twin below


def verify_proof_native(params, vk, proof: bytes, instances, flavour: str = "evm") -> bool:
    """the same check by the library's compiled host code (include/summa_prover.h: sp_verify_proof; csrc/verifier_abi.hip):
    no interpreter in the loop and the GIL released, so that the re-verification of every proof (`create_proof_checked`)
    does not serialise the proofs in flight of a batch"""
    if flavour not in ("evm", "blake2b"):
        raise ValueError("flavour: evm or blake2b")
    if any(not 0 <= int(v) < R for v in instances) or len(params.g2) != 128 or len(params.s_g2) != 128:
        return False
    cached = getattr(vk, "_native_view", None)
    if cached is None:
        pts = lambda comms: np.frombuffer(b"".join(_g1_bytes(tuple(c)) for c in comms), dtype=np.uint8).copy()
        cached = (np.frombuffer(int(vk.transcript_repr).to_bytes(32, "big"), dtype=np.uint8).copy(), pts(vk.fixed_comms),
                  pts(vk.permutation_comms))
        try:
            vk._native_view = cached
        except AttributeError:
            pass
    digest, fixed, perm = cached
    g2 = np.frombuffer(bytes(params.g2), dtype=np.uint8).copy()
    s_g2 = np.frombuffer(bytes(params.s_g2), dtype=np.uint8).copy()
    raw = np.frombuffer(bytes(proof), dtype=np.uint8).copy() if len(proof) else np.zeros(1, dtype=np.uint8)
    inst = np.frombuffer(b"".join(_fr_bytes(int(v)) for v in instances), dtype=np.uint8).copy() if len(instances) else np.zeros(1, dtype=np.uint8)
    ok = C.c_int(0)
    L = ffi.prover_lib()
    rc = L.sp_verify_proof(C.c_uint32(vk.k), C.c_uint32(vk.n_currencies), ffi.ptr(digest), ffi.ptr(fixed), ffi.ptr(perm), ffi.ptr(g2),
                           ffi.ptr(s_g2), ffi.ptr(raw), C.c_size_t(len(proof)), ffi.ptr(inst), C.c_uint32(len(instances)),
                           C.c_int(0 if flavour == "evm" else 1), C.byref(ok))
    if rc != 0:
        raise ffi.SummaGpuError(rc, L.sp_verify_last_error().decode())
    return ok.value == 1


def verify_proof(params, vk, proof: bytes, instances, flavour: str = "evm", driver: str | None = None) -> bool:
    """params: ParamsKZG with g2 / s_g2 (the verifier params); vk: api.VerifyingKey; instances: the instance column's
    values (integers < r).  True iff the proof is accepted.  A proof that cannot be checked -- malformed bytes, a point
    off the curve, a challenge that lands on the domain so that a denominator vanishes -- is a rejected proof, not an
    exception (upstream's `verify_proof` returns Err for all of these); a missing library or GPU still raises.
    driver: "native" (default: the library's compiled verifier) or "python" (this file's twin of it, same steps)."""
    if (driver or DRIVER) == "native":
        return verify_proof_native(params, vk, proof, instances, flavour)
    if any(not 0 <= int(v) < R for v in instances) or len(params.g2) != 128 or len(params.s_g2) != 128:
        return False
    if len(instances) > (1 << vk.k) - (M.BLINDING_FACTORS + 1):      # halo2: Error::InstanceTooLarge
        return False
    try:
        return _verify(params, vk, proof, instances, flavour)
    except (ValueError, ZeroDivisionError):
        return False
    except ffi.SummaGpuError as ex:
        if ex.code == -1:          # SG_ERR_INVALID: the data (an unreduced coordinate, a point off the curve), not the device
            return False
        raise


def _verify(params, vk, proof: bytes, instances, flavour: str) -> bool:
    k = vk.k
    try:
        rd = _Reader(bytes(proof), flavour)
        tr = rd.tr
        tr.common_scalar(vk.transcript_repr)
        for v in instances:
            tr.common_scalar(int(v))
        comms, ch = {}, {}
        for j in range(M.NUM_ADVICE):
            comms[("a", j)] = rd.read_point()
        ch["theta"] = tr.squeeze_challenge()
        comms[("pin", 0)], comms[("ptab", 0)] = rd.read_point(), rd.read_point()
        ch["beta"], ch["gamma"] = tr.squeeze_challenge(), tr.squeeze_challenge()
        for key in (("z", 0), ("z", 1), ("lz", 0), ("random", 0)):
            comms[key] = rd.read_point()
        ch["y"] = tr.squeeze_challenge()
        pieces = [rd.read_point() for _ in range(M.DEGREE - 1)]
        ch["x"] = tr.squeeze_challenge()
        evals = {key: rd.read_scalar() for key in P.EVAL_ORDER}
        zeta, nu = tr.squeeze_challenge(), tr.squeeze_challenge()
        w = rd.read_point()
        mu = tr.squeeze_challenge()
        w2 = rd.read_point()
        if rd.pos != len(rd.proof):
            return False
    except ValueError:
        return False
    x = ch["x"]
    lag = _lagrange(x, k, [int(v) for v in instances])
    h_eval = _expected_h_eval(evals, ch, lag, vk.n_currencies)
    for j, c in enumerate(vk.fixed_comms):
        comms[("f", j)] = tuple(c)
    for j, c in enumerate(vk.permutation_comms):
        comms[("sigma", j)] = tuple(c)

    # SHPLONK: per rotation set, the zeta-combination of its polynomials, interpolated through the claimed values and
    # evaluated at mu; sets weighted by nu^i * Z_{T \ S_i}(mu) / Z_{T \ S_0}(mu)
    n = 1 << k
    omega = pow(P.ROOT_OF_UNITY, 1 << (28 - k), R)
    point = lambda rot: x * pow(omega, rot % n, R) % R
    sets = P.ROTATION_SETS
    all_rots = sorted({r for rots, _ in sets for r in rots})
    mu_minus = {r: (mu - point(r)) % R for r in all_rots}
    outside = []
    for rots, _ in sets:
        d = 1
        for r in all_rots:
            if r not in rots:
                d = d * mu_minus[r] % R
        outside.append(d)
    norm0 = _inv(outside[0])
    z_s0 = 1
    for r in sets[0][0]:
        z_s0 = z_s0 * mu_minus[r] % R
    coeff = {}                     # commitment key -> scalar of the final multi-scalar multiplication
    r_eval, nu_pow = 0, 1
    for (rots, keys), d in zip(sets, outside):
        weights = []               # barycentric weights of the set's points, evaluated at mu
        for r in rots:
            den = mu_minus[r]
            for r2 in rots:
                if r2 != r:
                    den = den * (point(r) - point(r2)) % R
            weights.append(_inv(den))
        total = _inv(sum(weights) % R)
        scale = nu_pow * d % R * norm0 % R
        zeta_pow = 1
        for key in keys:
            value_at = lambda rot: h_eval if key == ("h", None) else evals[(key[0], key[1], rot)]
            at_mu = sum(wt * value_at(r) for wt, r in zip(weights, rots)) % R * total % R
            r_eval = (r_eval + scale * zeta_pow % R * at_mu) % R
            coeff[key] = (coeff.get(key, 0) + scale * zeta_pow) % R
            zeta_pow = zeta_pow * zeta % R
        nu_pow = nu_pow * nu % R
    points, scalars = [], []
    for key, c in coeff.items():
        if key == ("h", None):     # h(X) = sum_j x^(n j) h_j(X)
            for j, piece in enumerate(pieces):
                points.append(piece)
                scalars.append(c * pow(lag["x_n"], j, R) % R)
        else:
            points.append(comms[key])
            scalars.append(c)
    points += [(1, 2), w, w2]
    scalars += [(-r_eval) % R, (-z_s0) % R, mu]
    lhs = best_multiexp(np.frombuffer(b"".join(_fr_bytes(s) for s in scalars), dtype=np.uint8),
                        np.frombuffer(b"".join(_g1_bytes(p) for p in points), dtype=np.uint8))
    neg_w2 = _g1_bytes((w2[0], (-w2[1]) % Q))
    g1 = np.concatenate([lhs, np.frombuffer(neg_w2, dtype=np.uint8)])
    g2 = np.frombuffer(bytes(params.g2) + bytes(params.s_g2), dtype=np.uint8).copy()
    ok = C.c_int(0)
    ffi.check(ffi.lib().sg_pairing_check(ffi.ptr(g1), ffi.ptr(g2), C.c_size_t(2), C.byref(ok)))
    return ok.value == 1
