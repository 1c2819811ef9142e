"""Poseidon parameters for BN254 Fr, width t = 2 (rate 1), R_F = 8, R_P = 56, alpha = 5 --
regenerated from the PUBLISHED parameter-generation algorithm of the Poseidon authors
(Grain LFSR in self-shrinking mode; "generate_parameters_grain" of the Hades/Poseidon
reference implementation, also the origin of halo2_gadgets' constants).  TEST INFRASTRUCTURE
and input data for the witness-side kernels: the numbers are public protocol constants.

The reference pins them in zk_prover/src/chips/poseidon/poseidon_params.rs (ROUND_CONSTANTS
[64][2], MDS, MDS_INV), produced by its circuit_parameters_gen/generate_params.py with the
arguments `1 0 254 2 8 56 <r>`; tests/golden/make_fixtures.py checks this generator's output
against that file when the reference is present.
"""
from __future__ import annotations

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _grain_bits(field: int, sbox: int, n: int, t: int, r_f: int, r_p: int):
    def bits(v, w):
        return [int(c) for c in bin(v)[2:].zfill(w)]
    state = bits(field, 2) + bits(sbox, 4) + bits(n, 12) + bits(t, 12) + bits(r_f, 10) + bits(r_p, 10) + [1] * 30
    assert len(state) == 80

    def step():
        b = state[62] ^ state[51] ^ state[38] ^ state[23] ^ state[13] ^ state[0]
        state.pop(0)
        state.append(b)
        return b
    for _ in range(160):
        step()
    while True:
        # self-shrinking: read pairs, emit the second bit of a pair whose first bit is 1
        b = step()
        while b == 0:
            step()
            b = step()
        yield step()


def generate(t: int = 2, r_f: int = 8, r_p: int = 56, n: int = 254, p: int = R):
    """-> (round_constants[(r_f + r_p)][t], mds[t][t], mds_inv[t][t]) as Python ints"""
    gen = _grain_bits(1, 0, n, t, r_f, r_p)

    def rand_bits(k):
        v = 0
        for _ in range(k):
            v = (v << 1) | next(gen)
        return v
    rc = []
    for _ in range((r_f + r_p) * t):
        v = rand_bits(n)
        while v >= p:
            v = rand_bits(n)
        rc.append(v)
    rcs = [rc[i * t:(i + 1) * t] for i in range(r_f + r_p)]
    # Cauchy matrix M[i][j] = 1 / (x_i + y_j) from 2t distinct sampled field elements
    while True:
        vals = [rand_bits(n) % p for _ in range(2 * t)]
        while len(set(vals)) != 2 * t:
            vals = [rand_bits(n) % p for _ in range(2 * t)]
        xs, ys = vals[:t], vals[t:]
        if any((x + y) % p == 0 for x in xs for y in ys):
            continue
        mds = [[pow((x + y) % p, -1, p) for y in ys] for x in xs]
        break
    assert t == 2, "inverse below is written for 2 x 2"
    det = (mds[0][0] * mds[1][1] - mds[0][1] * mds[1][0]) % p
    di = pow(det, -1, p)
    inv = [[mds[1][1] * di % p, (-mds[0][1]) * di % p], [(-mds[1][0]) * di % p, mds[0][0] * di % p]]
    return rcs, mds, inv
