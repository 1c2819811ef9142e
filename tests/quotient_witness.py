"""Valid permutation / lookup witnesses for the quotient-numerator tests (test infrastructure).

A correct set of constraint evaluations has one observable property that does not need the
reference prover: with a witness that satisfies the argument, the folded numerator vanishes on
the whole 2^k domain, so numerator / (X^n - 1) is a polynomial of the expected degree and the
top coefficients of `extended_to_coeff` are all zero; with a tampered witness they are not.
The builders below produce such witnesses the way halo2 lays them out (usable rows
0..u-1, row u = l_last, rows above = blinding)."""
import numpy as np

from oracle import oracle as O
from oracle import pyref as P


def fr_np(values):
    return np.frombuffer(P.frs_to_bytes(values), dtype=np.uint8).copy()


def selectors(k, blinding):
    n = 1 << k
    u = n - blinding - 1
    l0 = [0] * n
    l0[0] = 1
    l_last = [0] * n
    l_last[u] = 1
    l_active = [1 if i < u else 0 for i in range(n)]
    return u, fr_np(l0), fr_np(l_last), fr_np(l_active)


def permutation_witness(k, ncols, chunk_len, blinding, seed, beta, gamma):
    """-> (cols, sigmas, zs): Lagrange-basis columns (numpy, Montgomery bytes)"""
    n = 1 << k
    u = n - blinding - 1
    rng = np.random.default_rng(seed)
    w = P.omega_for(k)
    wp = [1] * n
    for i in range(1, n):
        wp[i] = wp[i - 1] * w % P.R
    dp = [pow(P.DELTA, c, P.R) for c in range(ncols)]
    # copy classes over the usable cells; cells of one class form one cycle of the permutation
    nclass = max(2, (ncols * u) // 3)
    cls = rng.integers(0, nclass, size=(ncols, u))
    class_val = P.random_fr(seed + 1, nclass)
    vals = [[class_val[cls[c][i]] if i < u else 0 for i in range(n)] for c in range(ncols)]
    blind = P.random_fr(seed + 2, ncols * (n - u))
    for c in range(ncols):
        for i in range(u, n):
            vals[c][i] = blind[c * (n - u) + i - u]
    sigma = [[dp[c] * wp[i] % P.R for i in range(n)] for c in range(ncols)]
    members = {}
    for c in range(ncols):
        for i in range(u):
            members.setdefault(int(cls[c][i]), []).append((c, i))
    for cells in members.values():
        for j, (c, i) in enumerate(cells):
            c2, i2 = cells[(j + 1) % len(cells)]
            sigma[c][i] = dp[c2] * wp[i2] % P.R
    cols = [fr_np(v) for v in vals]
    sigmas = [fr_np(s) for s in sigma]
    zs = []
    z0 = None
    zblind = P.random_fr(seed + 3, (ncols + 1) * (n - u))
    for s, start in enumerate(range(0, ncols, chunk_len)):
        ch = slice(start, min(start + chunk_len, ncols))
        z = O.permutation_product(cols[ch], sigmas[ch], beta, gamma, fr_np([pow(P.DELTA, start, P.R)]), k, z0)
        z0 = z[32 * u:32 * u + 32].copy()
        z[32 * (u + 1):] = fr_np(zblind[s * (n - u):s * (n - u) + n - u - 1])
        zs.append(z)
    return cols, sigmas, zs


def lookup_witness(k, blinding, seed, beta, gamma):
    """-> (a, s, a', s', z) Lagrange-basis columns"""
    n = 1 << k
    u = n - blinding - 1
    rng = np.random.default_rng(seed)
    table = P.random_fr(seed + 1, u)                     # distinct with overwhelming probability
    a = [table[j] for j in rng.integers(0, max(1, u // 2), size=u)]
    ap = sorted(a)
    used = {}
    sp = [None] * u
    for i in range(u):
        if i == 0 or ap[i] != ap[i - 1]:
            sp[i] = ap[i]
            used[ap[i]] = True
    rest = [t for t in table if t not in used]
    for i in range(u):
        if sp[i] is None:
            sp[i] = rest.pop()
    assert not rest
    blind = P.random_fr(seed + 2, 5 * (n - u))
    m = n - u
    full = [fr_np(col + blind[j * m:(j + 1) * m]) for j, col in enumerate([a, table, ap, sp])]
    z = O.lookup_product(full[0], full[1], full[2], full[3], beta, gamma)
    z[32 * (u + 1):] = fr_np(blind[4 * m:4 * m + m - 1])
    return (*full, z)


def top_coefficients_zero(coeffs, first_zero):
    """coeffs: full 2^ext_k coefficient vector (numpy bytes)"""
    return not coeffs[32 * first_zero:].any()


def verifier_numerator(ev, beta, gamma, y, x, chunk_len):
    """The constraint equation a halo2 verifier evaluates at the challenge point x, with Python integers, in the
    order the reference's generated verifier folds it (contracts/src/InclusionVerifier.sol:903-997: quotient
    numerator = ((...(gate) * y + perm_0) * y + ...) ; custom gates first, then permutation, then lookup).
    ev: dict of evaluations (ints) -- see test_quotient_pipeline_satisfying_witness."""
    R = P.R
    acc = 0
    fold = lambda acc, term: (acc * y + term) % R
    # custom gate  q (ga * gb(omega x) - gc)
    acc = fold(acc, ev["gq"] * (ev["ga"] * ev["gb_next"] - ev["gc"]))
    # permutation argument
    zs, zs_next, zs_last = ev["z"], ev["z_next"], ev["z_last"]
    acc = fold(acc, ev["l0"] * (1 - zs[0]))
    acc = fold(acc, ev["l_last"] * (zs[-1] * zs[-1] - zs[-1]))
    for s in range(1, len(zs)):
        acc = fold(acc, ev["l0"] * (zs[s] - zs_last[s - 1]))
    delta_pow = 1
    col = 0
    ncols = len(ev["cols"])
    for s in range(len(zs)):
        left, right = zs_next[s], zs[s]
        for _ in range(min(chunk_len, ncols - col)):
            left = left * (ev["cols"][col] + beta * ev["sigma"][col] + gamma) % R
            right = right * (ev["cols"][col] + beta * delta_pow * x + gamma) % R
            delta_pow = delta_pow * P.DELTA % R
            col += 1
        acc = fold(acc, ev["l_active"] * (left - right))
    # lookup argument
    z, zn = ev["lz"], ev["lz_next"]
    acc = fold(acc, ev["l0"] * (1 - z))
    acc = fold(acc, ev["l_last"] * (z * z - z))
    acc = fold(acc, ev["l_active"] * (zn * (ev["ap"] + beta) * (ev["sp"] + gamma) - z * (ev["a"] + beta) * (ev["s"] + gamma)))
    acc = fold(acc, ev["l0"] * (ev["ap"] - ev["sp"]))
    acc = fold(acc, ev["l_active"] * (ev["ap"] - ev["sp"]) * (ev["ap"] - ev["ap_prev"]))
    return acc % R
