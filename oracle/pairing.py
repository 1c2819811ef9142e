"""BN254 optimal-ate pairing check with Python integers (the EVM's precompile 0x08, EIP-197).

TEST INFRASTRUCTURE ONLY (see oracle/pyref.py): used by the restated verifier
(oracle/summa_verifier.py) and by the dev-time run of the reference's generated verifier
(oracle/yul_verifier_run.py); never by product code.

The reference ends `verifyProof` with `staticcall(gas(), 0x08, ...)` over two (G1, G2) pairs
[REF contracts/src/InclusionVerifier.sol:185-202, 1404-1410]; on the Rust side the same check is
`halo2_proofs::poly::kzg::strategy::SingleStrategy` -> `halo2curves::bn256::multi_miller_loop`
(crate not under /root/reference, Cargo.lock:2273).  This file restates the published algorithm:
Fq12 = Fq[w] / (w^12 - 18 w^6 + 82), Fq2 embedded by i = w^6 - 9, the sextic twist (x, y) -> (x w^2, y w^3),
Miller loop over 6u + 2 with the two Frobenius correction steps, final exponentiation (q^12 - 1) / r.
Slow and obvious; pinned by bilinearity (tests/test_verifier_cpu.py) and by the reference's own shipped
proof K6 verifying through it.
"""
from __future__ import annotations

from .pyref import Q, R

ATE_LOOP_COUNT = 29793968203157093288  # 6u + 2, u = 4965661367192848881
LOG_ATE_LOOP_COUNT = 63


# ------------------------------------------------------------------ Fq12 as polynomials in w
def f12(coeffs):
    return [c % Q for c in coeffs] + [0] * (12 - len(coeffs))


F12_ONE = f12([1])
F12_ZERO = f12([0])


def f12_add(a, b):
    return [(x + y) % Q for x, y in zip(a, b)]


def f12_sub(a, b):
    return [(x - y) % Q for x, y in zip(a, b)]


def f12_neg(a):
    return [(-x) % Q for x in a]


def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for k in range(22, 11, -1):  # w^12 = 18 w^6 - 82
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return [x % Q for x in t[:12]]


def f12_scalar(a, s):
    return [(x * s) % Q for x in a]


def _poly_deg(p):
    d = len(p) - 1
    while d > 0 and p[d] == 0:
        d -= 1
    return d


def f12_inv(a):
    """Extended Euclid in Fq[w] against the modulus w^12 - 18 w^6 + 82."""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [82, 0, 0, 0, 0, 0, (-18) % Q, 0, 0, 0, 0, 0, 1]
    while _poly_deg(low):
        # r = high // low (polynomial quotient)
        dl, dh = _poly_deg(low), _poly_deg(high)
        r = [0] * 13
        tmp = list(high)
        inv_lead = pow(low[dl], -1, Q)
        for i in range(dh - dl, -1, -1):
            c = (tmp[dl + i] * inv_lead) % Q
            r[i] = c
            if c:
                for j in range(dl + 1):
                    tmp[i + j] = (tmp[i + j] - c * low[j]) % Q
        nm, new = list(hm), list(high)
        for i in range(13):
            if lm[i] or low[i]:
                for j in range(13 - i):
                    if r[j]:
                        nm[i + j] = (nm[i + j] - lm[i] * r[j]) % Q
                        new[i + j] = (new[i + j] - low[i] * r[j]) % Q
        lm, low, hm, high = nm, new, lm, low
    inv0 = pow(low[0], -1, Q)
    return [(c * inv0) % Q for c in lm[:12]]


def f12_pow(a, e: int):
    out, base = F12_ONE, a
    while e:
        if e & 1:
            out = f12_mul(out, base)
        base = f12_mul(base, base)
        e >>= 1
    return out


# ------------------------------------------------------------------ curve points with Fq12 coordinates (affine)
def _double(pt):
    x, y = pt
    m = f12_mul(f12_scalar(f12_mul(x, x), 3), f12_inv(f12_scalar(y, 2)))
    nx = f12_sub(f12_mul(m, m), f12_scalar(x, 2))
    ny = f12_sub(f12_mul(m, f12_sub(x, nx)), y)
    return nx, ny


def _add(p1, p2):
    if p1 is None or p2 is None:
        return p1 if p2 is None else p2
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        return _double(p1) if y1 == y2 else None
    m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    nx = f12_sub(f12_sub(f12_mul(m, m), x1), x2)
    ny = f12_sub(f12_mul(m, f12_sub(x1, nx)), y1)
    return nx, ny


def _line(p1, p2, t):
    """Line through p1, p2 evaluated at t."""
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    if y1 == y2:
        m = f12_mul(f12_scalar(f12_mul(x1, x1), 3), f12_inv(f12_scalar(y1, 2)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    return f12_sub(xt, x1)


_W2 = f12([0, 0, 1])
_W3 = f12([0, 0, 0, 1])


def twist(g2):
    """G2 point over Fq2 ((x0, x1), (y0, y1)) (x = x0 + x1 i) -> the curve over Fq12."""
    (x0, x1), (y0, y1) = g2
    nx = f12([(x0 - 9 * x1) % Q, 0, 0, 0, 0, 0, x1])
    ny = f12([(y0 - 9 * y1) % Q, 0, 0, 0, 0, 0, y1])
    return f12_mul(nx, _W2), f12_mul(ny, _W3)


def cast_g1(p):
    return f12([p[0]]), f12([p[1]])


def miller_loop(g2, g1):
    """Unreduced pairing value f_{6u+2,Q}(P) * corrections; None inputs (identity) give 1."""
    if g1 is None or g2 is None:
        return F12_ONE
    Qt, Pt = twist(g2), cast_g1(g1)
    Rt, f = Qt, F12_ONE
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = f12_mul(f12_mul(f, f), _line(Rt, Rt, Pt))
        Rt = _double(Rt)
        if ATE_LOOP_COUNT & (1 << i):
            f = f12_mul(f, _line(Rt, Qt, Pt))
            Rt = _add(Rt, Qt)
    q1 = (f12_pow(Qt[0], Q), f12_pow(Qt[1], Q))
    nq2 = (f12_pow(q1[0], Q), f12_neg(f12_pow(q1[1], Q)))
    f = f12_mul(f, _line(Rt, q1, Pt))
    Rt = _add(Rt, q1)
    f = f12_mul(f, _line(Rt, nq2, Pt))
    return f


def final_exponentiation(f):
    return f12_pow(f, (Q ** 12 - 1) // R)


def pairing(g2, g1):
    return final_exponentiation(miller_loop(g2, g1))


def g2_is_on_curve(g2) -> bool:
    """y^2 = x^3 + 3 / (9 + i) over Fq2."""
    if g2 is None:
        return True
    from .pyref import _f2_mul, _f2_add, _f2_inv
    x, y = g2
    b2 = _f2_mul((3, 0), _f2_inv((9, 1)))
    return _f2_mul(y, y) == _f2_add(_f2_mul(_f2_mul(x, x), x), b2)


def pairing_check(pairs) -> bool:
    """EIP-197: prod e(g1_i, g2_i) == 1.  Points are affine integer tuples or None for the identity."""
    f = F12_ONE
    for g1, g2 in pairs:
        f = f12_mul(f, miller_loop(g2, g1))
    return final_exponentiation(f) == F12_ONE
