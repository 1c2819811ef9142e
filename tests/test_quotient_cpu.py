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


def _py_eval_graph(graph, row, cols, chal, beta, gamma, theta, y, prev, n_ext, rot_scale):
    """big-int evaluation of one row of a GraphEvaluator program (kinds / ops numbered as in the C ABI)"""
    from oracle import pyref as P
    toi = lambda arr, i: P.fr_from_bytes(arr[32 * i:32 * i + 32].tobytes())
    inter = []

    def val(v):
        kind, index, rot = v
        if kind == 0:
            return toi(graph["constants"], index)
        if kind == 1:
            return inter[index]
        if kind in (2, 3, 4):
            return toi(cols[kind - 2][index], (row + graph["rotations"][rot] * rot_scale) % n_ext)
        if kind == 5:
            return toi(chal, index)
        return {6: beta, 7: gamma, 8: theta, 9: y, 10: prev}[kind]

    for cal in graph["calculations"]:
        op, a = cal[0], val(cal[1])
        if op == 0: r = a + val(cal[2])
        elif op == 1: r = a - val(cal[2])
        elif op == 2: r = a * val(cal[2])
        elif op == 3: r = a * a
        elif op == 4: r = 2 * a
        elif op == 5: r = -a
        elif op == 6:
            f, r = val(cal[2]), a
            for part in cal[3]:
                r = r * f + val(part)
        else: r = a
        inter.append(r % P.R)
    return inter[-1]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_gate_interpreter_vs_python_integers(seed):
    """pins orc_quotient_gates (the checker of the GPU interpreter) against a big-int evaluation"""
    from oracle import pyref as P
    rng = np.random.default_rng(seed)
    k, ext_k = 3, 5
    ne, scale = 1 << ext_k, 1 << (ext_k - k)
    cols = [[O.random_fr(50 * seed + 10 * j + i, ne) for i in range(2)] for j in range(3)]
    chal = O.random_fr(77, 2)
    consts = O.random_fr(78, 3)
    chs = [O.random_fr(80 + i, 1) for i in range(4)]
    calcs = []

    def src():
        kind = int(rng.integers(0, 11))
        if kind == 0: return (0, int(rng.integers(3)), 0)
        if kind == 1: return (1, int(rng.integers(len(calcs))), 0) if calcs else (9, 0, 0)
        if kind in (2, 3, 4): return (kind, int(rng.integers(2)), int(rng.integers(3)))
        if kind == 5: return (5, int(rng.integers(2)), 0)
        return (kind, 0, 0)

    for _ in range(40):
        op = int(rng.integers(0, 8))
        calcs.append((op, src(), src(), [src() for _ in range(int(rng.integers(0, 4)))]) if op == 6 else (op, src(), src()))
    graph = {"constants": consts, "rotations": [0, 1, -2], "calculations": calcs}
    start = O.random_fr(90, ne)
    got = O.quotient_gates(start, graph, cols[0], cols[1], cols[2], chal, *chs, k, ext_k)
    ints = [P.fr_from_bytes(c.tobytes()) for c in chs]
    for rowi in range(ne):
        want = _py_eval_graph(graph, rowi, cols, chal, *ints, P.fr_from_bytes(start[32 * rowi:32 * rowi + 32].tobytes()), ne, scale)
        assert P.fr_from_bytes(got[32 * rowi:32 * rowi + 32].tobytes()) == want


def test_threaded_row_loops_equal_the_serial_ones():
    """orc_set_quotient_threads (used by bench.py's proof-level CPU baseline): the three quotient blocks on 3 pthreads
    give the serial result, also with row counts that do not divide evenly"""
    k, ext_k = 9, 11
    ne = 1 << ext_k
    rnd = lambda s: O.random_fr(s, ne)
    beta, gamma, theta, y = (O.random_fr(300 + i, 1) for i in range(4))
    graph = {"constants": O.random_fr(310, 2), "rotations": [0, 1, -1],
             "calculations": [(2, (3, 0, 1), (2, 1, 2)), (0, (1, 0, 0), (0, 1, 0)), (3, (1, 1, 0), (0, 0, 0)),
                              (6, (10, 0, 0), (9, 0, 0), [(1, 2, 0), (3, 1, 0)])]}
    fixed, advice, inst = [rnd(320), rnd(321)], [rnd(322), rnd(323)], [rnd(324)]
    perm_args = ([rnd(330), rnd(331)], [rnd(332 + i) for i in range(5)], [rnd(340 + i) for i in range(5)], 3, rnd(350), rnd(351), rnd(352),
                 beta, gamma, y, k, ext_k, 6)
    look_args = (rnd(360), rnd(361), rnd(362), rnd(363), rnd(364), rnd(365), rnd(366), rnd(367), beta, gamma, y, k, ext_k)
    start = rnd(370)

    def run():
        return (O.quotient_gates(start, graph, fixed, advice, inst, O.random_fr(311, 1), beta, gamma, theta, y, k, ext_k),
                O.quotient_permutation(start, *perm_args), O.quotient_lookup(start, *look_args))
    try:
        serial = run()
        for threads in (3, 7):
            O.set_quotient_threads(threads)
            for a, b in zip(serial, run()):
                assert (a == b).all(), threads
    finally:
        O.set_quotient_threads(1)
