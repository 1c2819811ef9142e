"""Pure-Python big-integer twin of the BN254 MSM / NTT hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker.

The arithmetic of the path lives in third-party crates that are NOT present under
/root/reference (halo2_proofs 0.2.0 @ summa-dev/halo2#8386d6e, halo2curves 0.1.0;
zk_prover/Cargo.lock:2223-2276).  This file restates their *published* algorithms
 with Python integers, slowly and obviously,
so that the C restatement (bn254_oracle.c) and the HIP kernels have an independent
second implementation to be compared with.  It is pinned against the reference's own
artefacts: the SRS file backend/ptau/hermez-raw-11 (K1, K3), the verifier-contract
constants contracts/src/InclusionVerifier.sol:217-271 (K2, K4).

Reference call sites of the path: zk_prover/src/circuits/utils.rs:55,64,70,75,76,94-101.
"""
from __future__ import annotations

import struct

# --- field / curve constants (contracts/src/InclusionVerifier.sol:209-210) -------------
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # base field
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # scalar field
MONT = 1 << 256            # Montgomery radix of halo2curves' 4x64-bit limb representation
B_COEFF = 3                # y^2 = x^3 + 3
G1_GEN = (1, 2)
S_ADICITY = 28
# 2^28-th primitive root of unity of Fr (halo2curves `Fr::ROOT_OF_UNITY`; SURVEY.md K4,
# cross-checked against omega(k=11) in InclusionVerifier.sol:220)
ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
# cube root of unity used by halo2's EvaluationDomain as the extended-coset shift (`Fr::ZETA`)
ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
MULT_GEN = 7
DELTA = pow(MULT_GEN, 1 << S_ADICITY, R)


def omega_for(k: int) -> int:
    """Generator of the 2^k domain: ROOT_OF_UNITY^(2^(28-k)) (EvaluationDomain::new)."""
    assert 0 <= k <= S_ADICITY
    return pow(ROOT_OF_UNITY, 1 << (S_ADICITY - k), R)


# --- byte codecs (4 x u64 little-endian limbs, Montgomery form) ---------
def fr_to_bytes(x: int) -> bytes:
    return ((x * MONT) % R).to_bytes(32, "little")


def fr_from_bytes(b: bytes) -> int:
    return (int.from_bytes(b, "little") * pow(MONT, -1, R)) % R


def fq_to_bytes(x: int) -> bytes:
    return ((x * MONT) % Q).to_bytes(32, "little")


def fq_from_bytes(b: bytes) -> int:
    return (int.from_bytes(b, "little") * pow(MONT, -1, Q)) % Q


def g1_to_bytes(p) -> bytes:
    """Affine point -> 64 B (x||y Montgomery); identity (None) -> 64 zero bytes."""
    if p is None:
        return bytes(64)
    return fq_to_bytes(p[0]) + fq_to_bytes(p[1])


def g1_from_bytes(b: bytes):
    if b == bytes(64):
        return None
    return (fq_from_bytes(b[:32]), fq_from_bytes(b[32:64]))


def frs_to_bytes(xs) -> bytes:
    return b"".join(fr_to_bytes(x) for x in xs)


def frs_from_bytes(b: bytes):
    return [fr_from_bytes(b[i:i + 32]) for i in range(0, len(b), 32)]


# --- G1 arithmetic (affine, None = identity) --------------------------------------------
def g1_is_on_curve(p) -> bool:
    if p is None:
        return True
    x, y = p
    return (y * y - x * x * x - B_COEFF) % Q == 0


def g1_neg(p):
    return None if p is None else (p[0], (-p[1]) % Q)


def g1_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    y3 = (lam * (x1 - x3) - y1) % Q
    return (x3, y3)


# Jacobian helpers keep the pure-Python scalar multiplication tolerable.
def _jac_double(P):
    X, Y, Z = P
    if Z == 0:
        return P
    A = X * X % Q
    Bv = Y * Y % Q
    C = Bv * Bv % Q
    D = 2 * ((X + Bv) * (X + Bv) - A - C) % Q
    E = 3 * A % Q
    F = E * E % Q
    X3 = (F - 2 * D) % Q
    Y3 = (E * (D - X3) - 8 * C) % Q
    Z3 = 2 * Y * Z % Q
    return (X3, Y3, Z3)


def _jac_add_affine(P, q):
    if q is None:
        return P
    X1, Y1, Z1 = P
    x2, y2 = q
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % Q
    U2 = x2 * Z1Z1 % Q
    S2 = y2 * Z1 % Q * Z1Z1 % Q
    if U2 == X1:
        if S2 == Y1:
            return _jac_double(P)
        return (1, 1, 0)
    H = (U2 - X1) % Q
    HH = H * H % Q
    HHH = H * HH % Q
    r = (S2 - Y1) % Q
    V = X1 * HH % Q
    X3 = (r * r - HHH - 2 * V) % Q
    Y3 = (r * (V - X3) - Y1 * HHH) % Q
    Z3 = Z1 * H % Q
    return (X3, Y3, Z3)


def _jac_to_affine(P):
    X, Y, Z = P
    if Z == 0:
        return None
    zi = pow(Z, -1, Q)
    zi2 = zi * zi % Q
    return (X * zi2 % Q, Y * zi2 % Q * zi % Q)


def g1_mul(p, k: int):
    """k*p by left-to-right double-and-add; k is reduced mod r first."""
    k %= R
    if p is None or k == 0:
        return None
    acc = (1, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jac_double(acc)
        if bit == "1":
            acc = _jac_add_affine(acc, p)
    return _jac_to_affine(acc)


def msm_naive(scalars, points):
    """Definition of best_multiexp (M1): sum_i s_i * P_i."""
    assert len(scalars) == len(points)
    acc = None
    for s, p in zip(scalars, points):
        acc = g1_add(acc, g1_mul(p, s))
    return acc


def msm_pippenger(scalars, points, c: int | None = None):
    """Bucket method in the shape of halo2's multiexp_serial:
    unsigned c-bit digits of the canonical scalar, windows high->low, zero digits skipped,
    running-sum bucket reduction."""
    n = len(scalars)
    assert n == len(points)
    if c is None:
        if n < 4:
            c = 1
        elif n < 32:
            c = 3
        else:
            import math
            c = int(math.ceil(math.log(n)))
    segments = 256 // c + 1
    acc = (1, 1, 0)
    for seg in reversed(range(segments)):
        for _ in range(c):
            acc = _jac_double(acc)
        buckets = [(1, 1, 0)] * ((1 << c) - 1)
        for s, p in zip(scalars, points):
            d = ((s % R) >> (seg * c)) & ((1 << c) - 1)
            if d:
                buckets[d - 1] = _jac_add_affine(buckets[d - 1], p)
        running = None
        for b in reversed(buckets):
            running = g1_add(running, _jac_to_affine(b))
            acc_aff = g1_add(_jac_to_affine(acc), running)
            acc = (1, 1, 0) if acc_aff is None else (acc_aff[0], acc_aff[1], 1)
    return _jac_to_affine(acc)


# --- SRS container (ParamsKZG::read, RawBytes) -----------------------------
def parse_srs(raw: bytes):
    """k:u32 LE || g[2^k] || g_lagrange[2^k] || g2 || s_g2  (G1 64 B, G2 128 B)."""
    (k,) = struct.unpack_from("<I", raw, 0)
    n = 1 << k
    assert len(raw) == 4 + 2 * n * 64 + 2 * 128, "unexpected SRS length"
    g = raw[4:4 + 64 * n]
    gl = raw[4 + 64 * n:4 + 128 * n]
    g2 = raw[4 + 128 * n:4 + 128 * n + 128]
    s_g2 = raw[4 + 128 * n + 128:]
    return {"k": k, "n": n, "g": g, "g_lagrange": gl, "g2": g2, "s_g2": s_g2}


# --- NTT (best_fft, N1) and EvaluationDomain ops (N2-N4) ---------------------------------
def bitrev(x: int, bits: int) -> int:
    return int(bin(x)[2:].zfill(bits)[::-1], 2) if bits else 0


def ntt(a, omega: int, log_n: int):
    """Natural-order in/out forward DFT A[j] = sum_i a[i] omega^(ij) over Fr:
    bit-reverse permutation, then iterative radix-2 DIT (the serial shape of best_fft)."""
    n = 1 << log_n
    assert len(a) == n
    a = [x % R for x in a]
    for k in range(n):
        rk = bitrev(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    tw = [1] * max(n // 2, 1)
    for i in range(1, n // 2):
        tw[i] = tw[i - 1] * omega % R
    chunk, tchunk = 2, n // 2
    for _ in range(log_n):
        half = chunk // 2
        for s in range(0, n, chunk):
            for i in range(half):
                t = a[s + half + i] * tw[i * tchunk] % R
                u = a[s + i]
                a[s + i] = (u + t) % R
                a[s + half + i] = (u - t) % R
        chunk *= 2
        tchunk //= 2
    return a


def dft_naive(a, omega: int):
    n = len(a)
    return [sum(a[i] * pow(omega, i * j, R) for i in range(n)) % R for j in range(n)]


def intt(a, log_n: int):
    """EvaluationDomain::lagrange_to_coeff / ifft (N2): best_fft with omega^-1, then * n^-1."""
    w = omega_for(log_n)
    out = ntt(a, pow(w, -1, R), log_n)
    ninv = pow(1 << log_n, -1, R)
    return [x * ninv % R for x in out]


def coeff_to_extended(coeffs, k: int, ext_k: int):
    """EvaluationDomain::coeff_to_extended (N3): a[i] *= zeta^(i mod 3); zero-pad to
    2^ext_k; best_fft with omega_ext."""
    n = 1 << k
    assert len(coeffs) == n
    zp = [1, ZETA, ZETA * ZETA % R]
    a = [coeffs[i] * zp[i % 3] % R for i in range(n)] + [0] * ((1 << ext_k) - n)
    return ntt(a, omega_for(ext_k), ext_k)


def t_evaluations(k: int, ext_k: int):
    """1/(X^n - 1) on the coset zeta*<omega_ext>: period 2^(ext_k-k) table."""
    n = 1 << k
    w = omega_for(ext_k)
    out = []
    for i in range(1 << (ext_k - k)):
        x = ZETA * pow(w, i, R) % R
        out.append(pow(pow(x, n, R) - 1, -1, R))
    return out


def divide_by_vanishing_poly(ext, k: int, ext_k: int):
    t = t_evaluations(k, ext_k)
    return [v * t[i % len(t)] % R for i, v in enumerate(ext)]


def extended_to_coeff(ext, k: int, ext_k: int, quotient_degree: int | None = None):
    """EvaluationDomain::extended_to_coeff (N4): iNTT over the extended domain, undo the
    zeta powers (a[i] *= zeta^-(i mod 3)), truncate to n*(2^(ext_k-k) - ... ) per halo2:
    n * quotient_poly_degree (5n for this circuit); default keeps everything."""
    a = intt(ext, ext_k)
    zi = pow(ZETA, -1, R)
    zp = [1, zi, zi * zi % R]
    a = [a[i] * zp[i % 3] % R for i in range(len(a))]
    if quotient_degree is not None:
        a = a[: (1 << k) * quotient_degree]
    return a


# --- seeded inputs -----------------------------------------
MASK64 = (1 << 64) - 1
DEFAULT_SEED = 0x53554D4D41  # "SUMMA"


def splitmix64_stream(seed: int, count: int):
    """count successive outputs of SplitMix64 started at `seed`."""
    out = []
    s = seed & MASK64
    for _ in range(count):
        s = (s + 0x9E3779B97F4A7C15) & MASK64
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        out.append(z ^ (z >> 31))
    return out


def random_fr(seed: int, n: int):
    """n uniform Fr values: element i is built from SplitMix64 outputs 4i..4i+3 of the
    stream (limb 3 masked to 62 bits => 254-bit candidate) and accepted if < r; rejected
    candidates are replaced from a second stream seeded with seed ^ (i+1) -- a fixed,
    order-independent rule so numpy / C / Python generate identical vectors."""
    words = splitmix64_stream(seed, 4 * n)
    out = []
    for i in range(n):
        l = words[4 * i:4 * i + 4]
        v = l[0] | (l[1] << 64) | (l[2] << 128) | ((l[3] & ((1 << 62) - 1)) << 192)
        j = 0
        while v >= R:
            w = splitmix64_stream((seed ^ (i + 1)) + (j << 32), 4)
            v = w[0] | (w[1] << 64) | (w[2] << 128) | ((w[3] & ((1 << 62) - 1)) << 192)
            j += 1
        out.append(v)
    return out


# --- witness side: Poseidon Merkle sum tree ---------------------
# zk_prover/src/merkle_sum_tree/{entry.rs:15-27, node.rs:16-84, utils/build_tree.rs:5-78,
# utils/operation_helpers.rs:10-12, mst.rs:74-134}; Poseidon = halo2_gadgets' Pow5 sponge with
# the parameters of chips/poseidon/poseidon_spec.rs:14-37 (t = 2, rate 1, R_F = 8, R_P = 56).
def keccak256(data: bytes) -> bytes:
    """Keccak-256 (the pre-NIST padding 0x01 .. 0x80 that Ethereum uses), rate 136"""
    RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
          0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
          0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
          0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
          0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
    ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
    M = (1 << 64) - 1
    rol = lambda v, r: ((v << r) | (v >> (64 - r))) & M if r else v
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        for rnd in range(24):
            Cc = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
            D = [Cc[(x - 1) % 5] ^ rol(Cc[(x + 1) % 5], 1) for x in range(5)]
            A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
            B = [[0] * 5 for _ in range(5)]
            for x in range(5):
                for y in range(5):
                    B[y][(2 * x + 3 * y) % 5] = rol(A[x][y], ROT[x][y])
            A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
            A[0][0] ^= RC[rnd]
    out = b"".join(A[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


_POSEIDON = None


def _poseidon_params():
    global _POSEIDON
    if _POSEIDON is None:
        from . import poseidon_params
        _POSEIDON = poseidon_params.generate()
    return _POSEIDON


def poseidon_permute(state):
    rcs, mds, _ = _poseidon_params()
    s = list(state)
    def mix(s):
        return [(mds[0][0] * s[0] + mds[0][1] * s[1]) % R, (mds[1][0] * s[0] + mds[1][1] * s[1]) % R]
    rnd = 0
    for _ in range(4):
        s = mix([pow((s[i] + rcs[rnd][i]) % R, 5, R) for i in range(2)])
        rnd += 1
    for _ in range(56):
        s = [(s[i] + rcs[rnd][i]) % R for i in range(2)]
        s[0] = pow(s[0], 5, R)
        s = mix(s)
        rnd += 1
    for _ in range(4):
        s = mix([pow((s[i] + rcs[rnd][i]) % R, 5, R) for i in range(2)])
        rnd += 1
    return s


def poseidon_hash(inputs):
    """halo2_gadgets poseidon::Hash<Fr, Spec, ConstantLength<L>, 2, 1>: capacity element L * 2^64,
    one input absorbed per permutation (rate 1), output = state[0]"""
    state = [0, (len(inputs) << 64) % R]
    for m in inputs:
        state[0] = (state[0] + m) % R
        state = poseidon_permute(state)
    return state[0]


def mst_entry(username: str, balances):
    """Entry::new (entry.rs:15-27) + big_uint_to_fp: (keccak256(username) as BE integer mod r, balances mod r)"""
    return int.from_bytes(keccak256(username.encode()), "big") % R, [b % R for b in balances]


def mst_leaf(user_fr: int, balances):
    return poseidon_hash([user_fr] + list(balances))


def mst_middle(left, right):
    """nodes are (hash, balances)"""
    bal = [(a + b) % R for a, b in zip(left[1], right[1])]
    return poseidon_hash(bal + [left[0], right[0]]), bal


def mst_build(entries):
    """entries: [(user_fr, [balances])] already ordered; pads with zero entries to 2^depth
    (mst.rs:103-134); returns (root, levels) with nodes as (hash, balances)"""
    n = len(entries)
    depth = max(0, (n - 1).bit_length())
    nc = len(entries[0][1])
    entries = list(entries) + [(0, [0] * nc)] * ((1 << depth) - n)
    level = [(mst_leaf(u, b), list(b)) for u, b in entries]
    levels = [level]
    for _ in range(depth):
        level = [mst_middle(level[i], level[i + 1]) for i in range(0, len(level), 2)]
        levels.append(level)
    return levels[-1][0], levels


# --------------------------------------------------------------------------- G2 (verifier side of the SRS)
# BN254 G2: y^2 = x^3 + 3/(9+u) over Fq2 = Fq[u]/(u^2+1); only the group law is needed here (g2, s_g2 of
# ParamsKZG: SURVEY.md K1 -- the container's last 256 bytes).  Elements of Fq2 are pairs (c0, c1).
G2_GENERATOR = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
                 11559732032986387107991004021392285783925812861821192530917403151452391805634),
                (8495653923123431417604973247489272438418190587263600148770280649306958101930,
                 4082367875863433681332203403145435568316851327593401208105741076214120093531))


def _f2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def _f2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def _f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def _f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * n % Q, -a[1] * n % Q)


def g2_add(p, q):
    """affine addition on G2; None = identity"""
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if p[1] != q[1] or p[1] == (0, 0):
            return None
        lam = _f2_mul(_f2_mul((3, 0), _f2_mul(p[0], p[0])), _f2_inv(_f2_add(p[1], p[1])))
    else:
        lam = _f2_mul(_f2_sub(q[1], p[1]), _f2_inv(_f2_sub(q[0], p[0])))
    x3 = _f2_sub(_f2_sub(_f2_mul(lam, lam), p[0]), q[0])
    return (x3, _f2_sub(_f2_mul(lam, _f2_sub(p[0], x3)), p[1]))


def g2_mul(p, k: int):
    acc = None
    for bit in bin(k % R)[2:] if k % R else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, p)
    return acc


def g2_to_bytes(p) -> bytes:
    if p is None:
        return bytes(128)
    return b"".join(fq_to_bytes(c) for c in (p[0][0], p[0][1], p[1][0], p[1][1]))


# --------------------------------------------------------------------------- ChaCha20 (RFC 8439) and the field-element stream of sg_fr_random_dev
def chacha20_block(key: bytes, counter: int, nonce: bytes) -> bytes:
    """RFC 8439 section 2.3: 32-byte key, 32-bit block counter, 12-byte nonce -> 64 bytes"""
    M = 0xFFFFFFFF
    rotl = lambda v, c: ((v << c) & M) | (v >> (32 - c))
    init = list(struct.unpack("<4I", b"expand 32-byte k")) + list(struct.unpack("<8I", key)) + [counter & M] + list(struct.unpack("<3I", nonce))
    x = list(init)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M; x[b] = rotl(x[b] ^ x[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return struct.pack("<16I", *[(a + b) & M for a, b in zip(x, init)])


def chacha_field_elements(key: bytes, stream_id: int, n: int):
    """the n integers sg_fr_random_dev writes (as raw 32-byte little-endian values, no representation change):
    element i = first 32 bytes of block(counter = i, nonce = (attempt, stream_lo, stream_hi)), top two bits cleared,
    redrawn with attempt + 1 while >= r"""
    out = []
    for i in range(n):
        attempt = 0
        while True:
            nonce = struct.pack("<3I", attempt, stream_id & 0xFFFFFFFF, stream_id >> 32)
            v = int.from_bytes(chacha20_block(key, i, nonce)[:32], "little") & ((1 << 254) - 1)
            if v < R:
                out.append(v)
                break
            attempt += 1
    return out


# ------------------------------------------------------------------ Blake2b (RFC 7693), for the Blake2b transcript
# Independent of hashlib (the product uses the standard library's): pinned on the RFC's "abc" vector and against
# hashlib on random inputs / personalisations by tests/test_transcript_cpu.py.
_B2B_IV = [0x6A09E667F3BCC908, 0xBB67AE8584CAA73B, 0x3C6EF372FE94F82B, 0xA54FF53A5F1D36F1,
           0x510E527FADE682D1, 0x9B05688C2B3E6C1F, 0x1F83D9ABFB41BD6B, 0x5BE0CD19137E2179]
_B2B_SIGMA = [[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
              [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
              [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
              [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
              [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0]]


def _b2b_compress(h, block: bytes, t: int, last: bool):
    m = [int.from_bytes(block[8 * i:8 * i + 8], "little") for i in range(16)]
    v = list(h) + list(_B2B_IV)
    v[12] ^= t & MASK64
    v[13] ^= t >> 64
    if last:
        v[14] ^= MASK64
    rotr = lambda x, n: ((x >> n) | (x << (64 - n))) & MASK64

    def g(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & MASK64; v[d] = rotr(v[d] ^ v[a], 32)
        v[c] = (v[c] + v[d]) & MASK64; v[b] = rotr(v[b] ^ v[c], 24)
        v[a] = (v[a] + v[b] + y) & MASK64; v[d] = rotr(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & MASK64; v[b] = rotr(v[b] ^ v[c], 63)
    for rnd in range(12):
        s = _B2B_SIGMA[rnd % 10]
        g(0, 4, 8, 12, m[s[0]], m[s[1]]); g(1, 5, 9, 13, m[s[2]], m[s[3]])
        g(2, 6, 10, 14, m[s[4]], m[s[5]]); g(3, 7, 11, 15, m[s[6]], m[s[7]])
        g(0, 5, 10, 15, m[s[8]], m[s[9]]); g(1, 6, 11, 12, m[s[10]], m[s[11]])
        g(2, 7, 8, 13, m[s[12]], m[s[13]]); g(3, 4, 9, 14, m[s[14]], m[s[15]])
    return [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]


def blake2b(data: bytes, digest_size: int = 64, person: bytes = b"", salt: bytes = b"", key: bytes = b"") -> bytes:
    """unkeyed / keyed sequential Blake2b with salt and personalisation (parameter block of RFC 7693 section 2.5 with
    the salt / personal words of the BLAKE2 specification)"""
    assert 1 <= digest_size <= 64 and len(person) <= 16 and len(salt) <= 16 and len(key) <= 64
    param = bytes([digest_size, len(key), 1, 1]) + bytes(28) + salt.ljust(16, b"\0") + person.ljust(16, b"\0")
    h = [_B2B_IV[i] ^ int.from_bytes(param[8 * i:8 * i + 8], "little") for i in range(8)]
    if key:
        data = key.ljust(128, b"\0") + data
    t = 0
    while len(data) > 128:
        t += 128
        h = _b2b_compress(h, data[:128], t, False)
        data = data[128:]
    t += len(data)
    h = _b2b_compress(h, data.ljust(128, b"\0"), t, True)
    return b"".join(x.to_bytes(8, "little") for x in h)[:digest_size]
