"""Pins the oracle's restatement of the permutation / lookup blocks of `evaluate_h`
(oracle/bn254_oracle.c: orc_quotient_permutation, orc_quotient_lookup) through the property a
correct constraint system has: a satisfying witness makes the folded numerator divisible by
X^n - 1 (formulas: contracts/src/InclusionVerifier.sol:903-997)."""
import numpy as np
import pytest

import quotient_witness as W
from oracle import oracle as O


def _ext(col, k, ext_k):
    return O.coeff_to_extended(O.lagrange_to_coeff(col, k), k, ext_k)


def _quotient_coeffs(values, k, ext_k):
    return O.extended_to_coeff(O.divide_by_vanishing_poly(values, k, ext_k), k, ext_k)


@pytest.mark.parametrize("k,ncols,chunk_len", [(5, 3, 2), (6, 4, 2), (5, 1, 1), (6, 5, 1)])
def test_oracle_permutation_quotient_is_a_polynomial(k, ncols, chunk_len):
    blinding, ext_k, n = 5, k + 2, 1 << k
    beta, gamma, y = O.random_fr(21, 1), O.random_fr(22, 1), O.random_fr(23, 1)
    u, l0, l_last, l_active = W.selectors(k, blinding)
    cols, sigmas, zs = W.permutation_witness(k, ncols, chunk_len, blinding, 100 + k, beta, gamma)
    E = lambda c: _ext(c, k, ext_k)
    args = ([E(z) for z in zs], [E(c) for c in cols], [E(s) for s in sigmas], chunk_len, E(l0), E(l_last), E(l_active),
            beta, gamma, y, k, ext_k, blinding + 1)
    start = O.random_fr(24, 1 << ext_k)[: 32 << ext_k]
    zero = np.zeros(32 << ext_k, dtype=np.uint8)
    values = O.quotient_permutation(zero, *args)
    # numerator degree <= (chunk_len + 2)(n - 1)  =>  quotient degree <= (chunk_len + 1) n - chunk_len - 2
    first_zero = (chunk_len + 1) * n - chunk_len - 1
    assert W.top_coefficients_zero(_quotient_coeffs(values, k, ext_k), first_zero)
    assert values.any()
    # the fold is affine in the running value: f(start) - f(0) = start * y^terms
    nterms = 2 + (len(zs) - 1) + len(zs)
    yp = W.fr_np([1])
    for _ in range(nterms):
        yp = O.fr_mul(yp, y)
    shifted = O.quotient_permutation(start, *args)
    want = np.concatenate([O.fr_add(O.fr_mul(start[i:i + 32].copy(), yp), values[i:i + 32].copy())
                           for i in range(0, 32 << ext_k, 32)])
    assert (shifted == want).all()
    # negative control: break one copy constraint
    bad = cols[0].copy()
    bad[32:64] = O.fr_add(bad[32:64].copy(), W.fr_np([1]))
    args_bad = (args[0], [E(bad)] + args[1][1:], *args[2:])
    assert not W.top_coefficients_zero(_quotient_coeffs(O.quotient_permutation(zero, *args_bad), k, ext_k), first_zero)


@pytest.mark.parametrize("k", [5, 7])
def test_oracle_lookup_quotient_is_a_polynomial(k):
    blinding, ext_k, n = 5, k + 2, 1 << k
    beta, gamma, y = O.random_fr(31, 1), O.random_fr(32, 1), O.random_fr(33, 1)
    u, l0, l_last, l_active = W.selectors(k, blinding)
    a, s, ap, sp, z = W.lookup_witness(k, blinding, 200 + k, beta, gamma)
    E = lambda c: _ext(c, k, ext_k)
    zero = np.zeros(32 << ext_k, dtype=np.uint8)
    cols = [E(z), E(ap), E(sp), E(a), E(s)]
    values = O.quotient_lookup(zero, *cols, E(l0), E(l_last), E(l_active), beta, gamma, y, k, ext_k)
    first_zero = 3 * n - 3                      # numerator degree <= 4(n - 1)
    assert W.top_coefficients_zero(_quotient_coeffs(values, k, ext_k), first_zero)
    assert values.any()
    bad = ap.copy()                             # a' no longer a permutation of a
    bad[64:96] = O.fr_add(bad[64:96].copy(), W.fr_np([1]))
    cols[1] = E(bad)
    values = O.quotient_lookup(zero, *cols, E(l0), E(l_last), E(l_active), beta, gamma, y, k, ext_k)
    assert not W.top_coefficients_zero(_quotient_coeffs(values, k, ext_k), first_zero)
