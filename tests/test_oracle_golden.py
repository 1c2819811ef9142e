"""Pins the CPU oracle (oracle/bn254_oracle.c) and its big-integer twin (oracle/pyref.py)
against the reference's own artefacts and against each other.
CPU only."""
import numpy as np
import pytest

from conftest import fr_np, golden_bin, point_np
from oracle import oracle as O
from oracle import pyref as P


def test_k1_srs_layout(srs11):
    assert srs11["k"] == 11 and srs11["n"] == 2048
    assert P.g1_from_bytes(srs11["g"][:64]) == (1, 2)
    # raw bytes of g[0].x are R mod q => Montgomery convention
    assert int.from_bytes(srs11["g"][:32], "little") == P.MONT % P.Q
    for i in (0, 1, 2, 1000, 2047):
        assert O.g1_is_on_curve(srs11["g_np"][64 * i:64 * i + 64])
        assert O.g1_is_on_curve(srs11["gl_np"][64 * i:64 * i + 64])
        assert P.g1_is_on_curve(P.g1_from_bytes(srs11["g_lagrange"][64 * i:64 * i + 64]))


def test_k4_domain_constants(kat):
    assert int(kat["k"], 16) == 11
    assert P.omega_for(11) == int(kat["omega"], 16)
    assert pow(P.omega_for(11), -1, P.R) == int(kat["omega_inv"], 16)
    assert pow(2048, -1, P.R) == int(kat["n_inv"], 16)
    assert P.fr_from_bytes(O.omega(11).tobytes()) == int(kat["omega"], 16)
    assert P.fr_from_bytes(O.omega_inv(11).tobytes()) == int(kat["omega_inv"], 16)
    assert P.fr_from_bytes(O.n_inv(11).tobytes()) == int(kat["n_inv"], 16)
    assert P.DELTA == int(kat["delta"])
    assert pow(P.ZETA, 3, P.R) == 1 and P.ZETA != 1
    assert P.fr_from_bytes(O.zeta().tobytes()) == P.ZETA
    assert pow(P.ROOT_OF_UNITY, 1 << 28, P.R) == 1 and pow(P.ROOT_OF_UNITY, 1 << 27, P.R) != 1


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_k2_fixed_comm4_is_a_256_term_msm(srs11, kat, threads):
    """fixed_comms[4] = sum_{i<256} i * g_lagrange[i] (u8 lookup table column;
    zk_prover/src/circuits/traits.rs:35-52, InclusionVerifier.sol:246-247)."""
    want = point_np((int(kat["fixed_comms"][4][0], 16), int(kat["fixed_comms"][4][1], 16)))
    got = O.best_multiexp(fr_np(range(256)), srs11["gl_np"][:256 * 64], threads)
    assert (got == want).all()
    # whole column as halo2 commits it: 2048 scalars, zero beyond row 255
    sc = fr_np(list(range(256)) + [0] * (2048 - 256))
    assert (O.best_multiexp(sc, srs11["gl_np"], threads) == want).all()


def test_k2_pyref(srs11, kat):
    gl = [P.g1_from_bytes(srs11["g_lagrange"][64 * i:64 * i + 64]) for i in range(256)]
    want = (int(kat["fixed_comms"][4][0], 16), int(kat["fixed_comms"][4][1], 16))
    assert P.msm_naive(list(range(256)), gl) == want
    assert P.msm_pippenger(list(range(256)), gl) == want


def test_k3_lagrange_monomial_relation(srs11):
    """MSM([omega^(ik)]_i, g_lagrange) = g[k]; ties NTT constants to MSM over 2048 points."""
    w = O.omega(11)
    one = fr_np([1])
    assert (O.best_multiexp(O.fr_powers(one, 2048), srs11["gl_np"], 4) == srs11["g_np"][:64]).all()
    for k in (1, 2, 5):
        wk = fr_np([pow(P.omega_for(11), k, P.R)])
        got = O.best_multiexp(O.fr_powers(wk, 2048), srs11["gl_np"], 4)
        assert (got == srs11["g_np"][64 * k:64 * k + 64]).all()
    # and the converse: commit(iNTT(e_j)) over g = g_lagrange[j]
    for j in (0, 3, 2047):
        e = [0] * 2048
        e[j] = 1
        coeffs = O.lagrange_to_coeff(fr_np(e), 11, 2)
        assert (O.best_multiexp(coeffs, srs11["g_np"], 4) == srs11["gl_np"][64 * j:64 * j + 64]).all()
    del w


def test_field_ops_match_bigint():
    vals = P.random_fr(17, 24) + [0, 1, P.R - 1, P.R - 2, 2]
    for a in vals[:12]:
        for b in vals[12:]:
            A, B = fr_np([a]), fr_np([b])
            assert P.fr_from_bytes(O.fr_mul(A, B).tobytes()) == a * b % P.R
            assert P.fr_from_bytes(O.fr_add(A, B).tobytes()) == (a + b) % P.R
            assert P.fr_from_bytes(O.fr_sub(A, B).tobytes()) == (a - b) % P.R
    for a in vals[:6]:
        assert P.fr_from_bytes(O.fr_inv(fr_np([a])).tobytes()) == pow(a, -1, P.R)
    qa = [x % P.Q for x in P.random_fr(5, 8)] + [P.Q - 1]
    for a in qa:
        for b in qa:
            A = np.frombuffer(P.fq_to_bytes(a), dtype=np.uint8).copy()
            B = np.frombuffer(P.fq_to_bytes(b), dtype=np.uint8).copy()
            assert P.fq_from_bytes(O.fq_mul(A, B).tobytes()) == a * b % P.Q


def test_random_fr_generators_agree():
    n = 300
    a = O.random_fr(P.DEFAULT_SEED, n)
    assert P.frs_from_bytes(a.tobytes()) == P.random_fr(P.DEFAULT_SEED, n)
    from circuits_halo2_amd.utils import random_fr_canonical, to_montgomery_host
    c = random_fr_canonical(P.DEFAULT_SEED, n)
    assert [int.from_bytes(c[32 * i:32 * i + 32].tobytes(), "little") for i in range(n)] == P.random_fr(P.DEFAULT_SEED, n)
    assert (to_montgomery_host(c) == a).all()


def test_ntt_golden_k4_and_k11(kat):
    a = [int(x, 16) for x in kat["ntt_k4"]["in"]]
    want = [int(x, 16) for x in kat["ntt_k4"]["out"]]
    assert P.ntt(a, P.omega_for(4), 4) == want
    for th in (1, 2, 8):
        assert P.frs_from_bytes(O.best_fft(fr_np(a), O.omega(4), 4, th).tobytes()) == want
    x, y = golden_bin("ntt_k11_in.bin"), golden_bin("ntt_k11_out.bin")
    for th in (1, 8):
        assert (O.best_fft(x, O.omega(11), 11, th) == y).all()
    assert (O.lagrange_to_coeff(y, 11, 4) == x).all()


@pytest.mark.parametrize("k", [1, 2, 3, 5, 8])
def test_ntt_c_vs_bigint(k):
    a = O.random_fr(100 + k, 1 << k)
    ai = P.frs_from_bytes(a.tobytes())
    assert P.frs_from_bytes(O.best_fft(a, O.omega(k), k, 4).tobytes()) == P.ntt(ai, P.omega_for(k), k)
    assert P.frs_from_bytes(O.lagrange_to_coeff(a, k, 4).tobytes()) == P.intt(ai, k)
    if k <= 5:
        assert P.ntt(ai, P.omega_for(k), k) == P.dft_naive(ai, P.omega_for(k))


def test_ntt_closed_forms():
    k = 10
    n = 1 << k
    c = P.random_fr(1, 1)[0]
    imp = fr_np([c] + [0] * (n - 1))
    assert (O.best_fft(imp, O.omega(k), k, 2) == fr_np([c] * n)).all()
    const = fr_np([c] * n)
    assert (O.best_fft(const, O.omega(k), k, 2) == fr_np([c * n % P.R] + [0] * (n - 1))).all()


def test_extended_domain_golden(kat):
    a = [int(x, 16) for x in kat["coeff_to_extended_k4_e7"]["in"]]
    want = [int(x, 16) for x in kat["coeff_to_extended_k4_e7"]["out"]]
    ext = O.coeff_to_extended(fr_np(a), 4, 7, 2)
    assert P.frs_from_bytes(ext.tobytes()) == want
    # the extended evaluations are f(zeta * omega_ext^j)
    w = P.omega_for(7)
    for j in (0, 1, 77):
        x = P.ZETA * pow(w, j, P.R) % P.R
        assert want[j] == sum(c * pow(x, i, P.R) for i, c in enumerate(a)) % P.R
    back = P.frs_from_bytes(O.extended_to_coeff(ext, 4, 7, 2).tobytes())
    assert back[:16] == a and not any(back[16:])
    tev = [int(x, 16) for x in kat["t_evaluations_k4_e7"]]
    assert P.t_evaluations(4, 7) == tev
    d = P.frs_from_bytes(O.divide_by_vanishing_poly(ext, 4, 7).tobytes())
    assert d == [v * tev[i % 8] % P.R for i, v in enumerate(want)]


def test_msm_tau_golden(kat):
    """commit(f) over the synthetic SRS g[i] = tau^i G equals f(tau) G."""
    sc, bases = golden_bin("msm_tau_k10_scalars.bin"), golden_bin("msm_tau_k10_bases.bin")
    want = point_np(tuple(int(x, 16) for x in kat["msm_tau_k10"]["answer"]))
    for th in (1, 8):
        assert (O.best_multiexp(sc, bases, th) == want).all()
    sp = fr_np([int(x, 16) for x in kat["msm_tau_k10_sparse"]["scalars"]])
    want = point_np(tuple(int(x, 16) for x in kat["msm_tau_k10_sparse"]["answer"]))
    assert (O.best_multiexp(sp, bases, 3) == want).all()
    # fixed-base generator of the oracle reproduces the golden bases
    tau = fr_np([int(kat["msm_tau_k10"]["tau"], 16)])
    assert (O.fixed_base_mul(O.fr_powers(tau, 1024), 4) == bases).all()


def test_msm_edge_cases(srs11):
    gl = srs11["gl_np"]
    ident = np.zeros(64, dtype=np.uint8)
    assert (O.best_multiexp(np.zeros(0, np.uint8), np.zeros(0, np.uint8), 1) == ident).all()
    assert (O.best_multiexp(fr_np([0] * 8), gl[:8 * 64], 2) == ident).all()
    p = gl[:64]
    negp = point_np(P.g1_neg(P.g1_from_bytes(p.tobytes())))
    # P + (-P) = identity; (r-1) P = -P; repeated points double
    assert (O.best_multiexp(fr_np([1, 1]), np.concatenate([p, negp]), 1) == ident).all()
    assert (O.best_multiexp(fr_np([P.R - 1]), p, 1) == negp).all()
    two_p = point_np(P.g1_mul(P.g1_from_bytes(p.tobytes()), 2))
    assert (O.best_multiexp(fr_np([1, 1]), np.concatenate([p, p]), 1) == two_p).all()
    # identity bases are ignored
    assert (O.best_multiexp(fr_np([5, 1]), np.concatenate([ident, p]), 1) == p).all()
    # all-equal scalars and points (one bucket gets everything)
    n = 100
    want = point_np(P.g1_mul(P.g1_from_bytes(p.tobytes()), 7 * n))
    assert (O.best_multiexp(fr_np([7] * n), np.tile(p, n), 4) == want).all()
