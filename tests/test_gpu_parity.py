"""GPU parity tests: every case calls the HIP path through the C ABI (libsumma_gpu.so via the
Python mirror of halo2's interface) and compares bit-for-bit with the CPU oracle, the
committed golden vectors, or a size-independent property.  Run on the MI355X box with
`pytest -m gpu`."""
import numpy as np
import pytest

from conftest import fr_np, golden_bin, point_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    import circuits_halo2_amd as sg
    assert sg.lib().sg_device_count() >= 1
    # SG_PARAMS=msm.two_pass=2,msm.quad=2 ... reruns the whole suite on a non-default code path
    import os
    from circuits_halo2_amd import ffi
    for kv in filter(None, os.environ.get("SG_PARAMS", "").split(",")):
        name, val = kv.split("=")
        ffi.check(sg.lib().sg_init(0))
        ffi.check(sg.lib().sg_set_param(name.encode(), int(val)))
    return sg


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def P():
    from oracle import pyref
    return pyref


def P_R():
    from oracle import pyref
    return pyref.R


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ----------------------------------------------------------------------------- NTT (N1)
def test_ntt_golden_vectors(gpu, O, kat):
    a = fr_np([int(x, 16) for x in kat["ntt_k4"]["in"]])
    want = fr_np([int(x, 16) for x in kat["ntt_k4"]["out"]])
    assert (gpu.best_fft(a, O.omega(4), 4) == want).all()
    x, y = golden_bin("ntt_k11_in.bin"), golden_bin("ntt_k11_out.bin")
    assert (gpu.best_fft(x, O.omega(11), 11) == y).all()


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 14, 16, 17, 18, 20])
def test_ntt_matches_oracle(gpu, O, log_n):
    a = O.random_fr(1000 + log_n, 1 << log_n)
    w = O.omega(log_n)
    want = O.best_fft(a, w, log_n, O.ncpu())
    assert (gpu.best_fft(a, w, log_n) == want).all()
    # device-resident entry point (in place on a torch tensor)
    d = dev(a)
    gpu.best_fft(d, w, log_n)
    assert (d.cpu().numpy() == want).all()


def test_ntt_three_pass_plan(gpu, O):
    log_n = 21  # > 2 * max_multi_log: exercises the three-pass (six-step) factorisation
    a = O.random_fr(77, 1 << log_n)
    w = O.omega(log_n)
    assert (gpu.best_fft(a, w, log_n) == O.best_fft(a, w, log_n, O.ncpu())).all()


def test_ntt_arbitrary_root(gpu, O, P):
    """best_fft takes any omega of the right order, not only the domain generator"""
    log_n = 9
    w = fr_np([pow(P.omega_for(log_n), 5, P.R)])
    a = O.random_fr(5, 1 << log_n)
    assert (gpu.best_fft(a, w, log_n) == O.best_fft(a, w, log_n, 2)).all()


def test_ntt_closed_forms(gpu, O, P):
    k, n = 13, 1 << 13
    c = P.random_fr(1, 1)[0]
    assert (gpu.best_fft(fr_np([c] + [0] * (n - 1)), O.omega(k), k) == fr_np([c] * n)).all()
    assert (gpu.best_fft(fr_np([c] * n), O.omega(k), k) == fr_np([c * n % P.R] + [0] * (n - 1))).all()


@pytest.mark.parametrize("log_n", [17, 22])
def test_ntt_roundtrip_full_size(gpu, O, log_n):
    """BASELINE config 3: iNTT(NTT(a)) == a bit-exactly, device resident"""
    dom = gpu.EvaluationDomain(2, log_n)
    a = O.random_fr(0x53554D4D42, 1 << log_n)
    d = dev(a)
    gpu.best_fft(d, dom.get_omega(), log_n)
    fwd = d.cpu().numpy().copy()
    assert not (fwd == a).all()
    dom.lagrange_to_coeff(d)
    assert (d.cpu().numpy() == a).all()
    # spot-check the forward transform against direct evaluation at three points
    for j in (0, 1, (1 << log_n) - 1):
        x = O.fr_powers(dom.get_omega(), j + 1)[32 * j:32 * j + 32].copy()
        assert (O.fr_eval_poly(a, x) == fwd[32 * j:32 * j + 32]).all()


def test_ntt_linearity(gpu, O):
    log_n = 15
    a, b = O.random_fr(1, 1 << log_n), O.random_fr(2, 1 << log_n)
    w = O.omega(log_n)
    s = np.concatenate([O.fr_add(a[i:i + 32].copy(), b[i:i + 32].copy()) for i in range(0, 32 * 64, 32)])
    fa, fb = gpu.best_fft(a, w, log_n), gpu.best_fft(b, w, log_n)
    ab = a.copy()
    for i in range(0, a.size, 32):
        pass
    # NTT(a) + NTT(b) == NTT(a + b) on the first 64 outputs needs the full sum; use the oracle
    # only for the element-wise additions
    import ctypes as C
    tot = np.zeros_like(a)
    lib = O.lib()
    for i in range(0, a.size, 32):
        lib.orc_fr_add(C.c_void_p(a.ctypes.data + i), C.c_void_p(b.ctypes.data + i), C.c_void_p(tot.ctypes.data + i))
    fs = gpu.best_fft(tot, w, log_n)
    chk = np.zeros(32 * 64, dtype=np.uint8)
    for i in range(0, 32 * 64, 32):
        lib.orc_fr_add(C.c_void_p(fa.ctypes.data + i), C.c_void_p(fb.ctypes.data + i), C.c_void_p(chk.ctypes.data + i))
    assert (fs[:32 * 64] == chk).all()
    del s, ab


# ----------------------------------------------------------------------------- N2-N4
@pytest.mark.parametrize("k", [1, 4, 9, 11, 12, 15])
def test_lagrange_to_coeff(gpu, O, k):
    a = O.random_fr(200 + k, 1 << k)
    dom = gpu.EvaluationDomain(6, k)
    want = O.lagrange_to_coeff(a, k, O.ncpu())
    assert (dom.lagrange_to_coeff(a) == want).all()
    assert (dom.get_omega() == O.omega(k)).all()
    assert (dom.get_omega_inv() == O.omega_inv(k)).all()
    assert (dom.ifft_divisor() == O.n_inv(k)).all()
    # explicit-constant form: EvaluationDomain::ifft(a, omega_inv, log_n, divisor)
    import ctypes as C
    from circuits_halo2_amd import ffi
    buf = a.copy()
    ffi.check(ffi.lib().sg_intt_fr(ffi.ptr(buf), ffi.ptr(O.omega_inv(k)), ffi.ptr(O.n_inv(k)), C.c_uint32(k)))
    assert (buf == want).all()


def test_domain_golden_k11(gpu, kat, P):
    dom = gpu.EvaluationDomain(6, 11)
    assert dom.extended_k == 14 and dom.quotient_poly_degree == 5
    assert P.fr_from_bytes(dom.get_omega().tobytes()) == int(kat["omega"], 16)
    assert P.fr_from_bytes(dom.get_omega_inv().tobytes()) == int(kat["omega_inv"], 16)
    assert P.fr_from_bytes(dom.ifft_divisor().tobytes()) == int(kat["n_inv"], 16)


def test_extended_domain_golden(gpu, kat):
    a = fr_np([int(x, 16) for x in kat["coeff_to_extended_k4_e7"]["in"]])
    want = fr_np([int(x, 16) for x in kat["coeff_to_extended_k4_e7"]["out"]])
    dom = gpu.EvaluationDomain(6, 4)
    assert dom.extended_k == 7
    assert (dom.coeff_to_extended(a) == want).all()
    tev = [int(x, 16) for x in kat["t_evaluations_k4_e7"]]
    ones = fr_np([1] * 128)
    assert (dom.divide_by_vanishing_poly(ones) == fr_np([tev[i % 8] for i in range(128)])).all()


@pytest.mark.parametrize("k", [3, 8, 11, 13, 17])
def test_extended_domain_vs_oracle(gpu, O, k):
    dom = gpu.EvaluationDomain(6, k)
    ek = dom.extended_k
    a = O.random_fr(300 + k, 1 << k)
    want = O.coeff_to_extended(a, k, ek, O.ncpu())
    got = dom.coeff_to_extended(a)
    assert (got == want).all()
    d = dom.coeff_to_extended(dev(a))
    assert (d.cpu().numpy() == want).all()
    assert (dom.divide_by_vanishing_poly(got) == O.divide_by_vanishing_poly(want, k, ek)).all()
    back = dom.extended_to_coeff(got)
    full = O.extended_to_coeff(want, k, ek, O.ncpu())
    assert back.size == 32 * 5 * (1 << k)
    assert (back == full[:back.size]).all()
    assert (back[:a.size] == a).all() and not back[a.size:].any()
    # a degree-5n polynomial survives the round trip through the extended domain
    q = O.random_fr(400 + k, 5 << k)
    qpad = np.concatenate([q, np.zeros((32 << ek) - q.size, dtype=np.uint8)])
    ext = O.best_fft(_coset_scale(O, qpad), O.omega(ek), ek, O.ncpu())
    assert (dom.extended_to_coeff(ext) == q).all()


def _coset_scale(O, a):
    import ctypes as C
    out = a.copy()
    z = O.zeta()
    z2 = O.fr_mul(z, z)
    lib = O.lib()
    for i in range(0, a.size // 32):
        m = i % 3
        if m:
            src = z if m == 1 else z2
            lib.orc_fr_mul(C.c_void_p(out.ctypes.data + 32 * i), C.c_void_p(src.ctypes.data),
                           C.c_void_p(out.ctypes.data + 32 * i))
    return out


# ----------------------------------------------------------------------------- MSM (M1)
def test_msm_k2_fixed_comm4(gpu, srs11, kat):
    want = point_np((int(kat["fixed_comms"][4][0], 16), int(kat["fixed_comms"][4][1], 16)))
    assert (gpu.best_multiexp(fr_np(range(256)), srs11["gl_np"][:256 * 64]) == want).all()
    params = gpu.ParamsKZG.read(open(__import__("os").path.join(__import__("conftest").GOLDEN, "hermez-raw-11"), "rb"))
    assert params.k == 11
    col = fr_np(list(range(256)) + [0] * (2048 - 256))
    assert (params.commit_lagrange(col) == want).all()
    assert (params.commit_lagrange(dev(col)) == want).all()
    params.free()


def test_msm_k3_srs_relations(gpu, O, P, srs11):
    params = gpu.ParamsKZG(11, srs11["g_np"], srs11["gl_np"])
    one = fr_np([1])
    assert (params.commit_lagrange(O.fr_powers(one, 2048)) == srs11["g_np"][:64]).all()
    for k in (1, 7):
        wk = fr_np([pow(P.omega_for(11), k, P.R)])
        assert (params.commit_lagrange(O.fr_powers(wk, 2048)) == srs11["g_np"][64 * k:64 * k + 64]).all()
    dom = gpu.EvaluationDomain(6, 11)
    for j in (0, 5, 2047):
        e = [0] * 2048
        e[j] = 1
        assert (params.commit(dom.lagrange_to_coeff(fr_np(e))) == srs11["gl_np"][64 * j:64 * j + 64]).all()
    params.free()


def test_msm_tau_golden(gpu, kat):
    sc, bases = golden_bin("msm_tau_k10_scalars.bin"), golden_bin("msm_tau_k10_bases.bin")
    want = point_np(tuple(int(x, 16) for x in kat["msm_tau_k10"]["answer"]))
    assert (gpu.best_multiexp(sc, bases) == want).all()
    sp = fr_np([int(x, 16) for x in kat["msm_tau_k10_sparse"]["scalars"]])
    want = point_np(tuple(int(x, 16) for x in kat["msm_tau_k10_sparse"]["answer"]))
    assert (gpu.best_multiexp(sp, bases) == want).all()


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 100, 1000, 1 << 12, 5000, 1 << 14, 1 << 16])
def test_msm_matches_oracle(gpu, O, n):
    sc = O.random_fr(500 + n, n)
    bases = O.fixed_base_mul(O.random_fr(900 + n, n), O.ncpu())
    want = O.best_multiexp(sc, bases, O.ncpu())
    assert (gpu.best_multiexp(sc, bases) == want).all()
    assert (gpu.best_multiexp(dev(sc), dev(bases)) == want).all()


@pytest.mark.parametrize("c", [4, 5, 7, 9, 11, 12, 13, 14, 15, 16])
def test_msm_window_sizes(gpu, O, c):
    from circuits_halo2_amd import ffi
    n = 3000
    sc = O.random_fr(11, n)
    bases = O.fixed_base_mul(O.random_fr(12, n), O.ncpu())
    want = O.best_multiexp(sc, bases, O.ncpu())
    ffi.check(ffi.lib().sg_set_param(b"msm.window_bits", c))
    try:
        assert (gpu.best_multiexp(sc, bases) == want).all()
    finally:
        ffi.check(ffi.lib().sg_set_param(b"msm.window_bits", 0))


def test_msm_edge_cases(gpu, O, P, srs11):
    gl = srs11["gl_np"]
    ident = np.zeros(64, dtype=np.uint8)
    assert (gpu.best_multiexp(np.zeros(0, np.uint8), np.zeros(0, np.uint8)) == ident).all()
    assert (gpu.best_multiexp(fr_np([0] * 8), gl[:8 * 64]) == ident).all()
    p = gl[:64]
    pt = P.g1_from_bytes(p.tobytes())
    negp = point_np(P.g1_neg(pt))
    assert (gpu.best_multiexp(fr_np([1, 1]), np.concatenate([p, negp])) == ident).all()
    assert (gpu.best_multiexp(fr_np([P.R - 1]), p) == negp).all()
    assert (gpu.best_multiexp(fr_np([1, 1]), np.concatenate([p, p])) == point_np(P.g1_mul(pt, 2))).all()
    assert (gpu.best_multiexp(fr_np([5, 1]), np.concatenate([ident, p])) == p).all()
    # scalars at the top of the range and powers of two straddling window boundaries
    vals = [P.R - 1, P.R - 2, (P.R - 1) // 2, 1 << 253, (1 << 128) - 1, 1 << 16, (1 << 16) - 1, 1 << 15, 65537, 2]
    bases = gl[:64 * len(vals)]
    assert (gpu.best_multiexp(fr_np(vals), bases) == O.best_multiexp(fr_np(vals), bases, 1)).all()
    with pytest.raises(ValueError):
        gpu.best_multiexp(fr_np([1, 2]), p)


@pytest.mark.parametrize("n", [600, 5000, 1 << 15])
def test_msm_repeated_and_opposite_bases(gpu, O, P, n):
    """every special case of the group law inside the pipeline: the same point many times in one bucket
    (P + P in accumulate, equal partial sums in merge / reduction: the doubling fallbacks, also of the
    quad-cooperative addition), P and -P cancelling to the identity; generic, fixed-base and fused paths"""
    g = O.g1_generator()
    gi = P.g1_from_bytes(g.tobytes())
    neg = point_np(P.g1_neg(gi))
    rng = np.random.default_rng(n)
    small = [int(v) for v in rng.integers(0, 4, size=n)]                    # few distinct digits: deep equal buckets
    sc_small = fr_np(small)
    sc_rand = O.random_fr(3000 + n, n)
    same = np.tile(g, n)
    total = lambda sc: O.fixed_base_mul(O.fr_dot(sc, fr_np([1] * n)), 1)   # (sum s_i) * G
    assert (gpu.best_multiexp(sc_small, same) == total(sc_small)).all()
    assert (gpu.best_multiexp(sc_rand, same) == total(sc_rand)).all()
    assert (gpu.best_multiexp(np.tile(sc_rand[:32], n), same) == total(np.tile(sc_rand[:32], n))).all()
    # alternating P, -P with equal scalars in pairs: everything cancels
    alt = np.concatenate([g, neg] * (n // 2))
    pair_sc = np.repeat(O.random_fr(3100 + n, n // 2).reshape(-1, 32), 2, axis=0).reshape(-1)
    assert not gpu.best_multiexp(pair_sc, alt).any()
    k = max(1, (n - 1).bit_length())
    pad = (1 << k) - n
    params = gpu.ParamsKZG(k, np.concatenate([same, np.tile(g, pad)]), np.concatenate([alt, np.tile(g, pad)]))
    params.precompute()
    assert (params.commit(sc_small) == total(sc_small)).all()
    assert (params.commit(sc_rand) == total(sc_rand)).all()
    assert not params.commit_lagrange(pair_sc).any()
    got = params.commit_batch([dev(sc_small), dev(sc_rand), dev(sc_small)])
    assert (got[0] == total(sc_small)).all() and (got[1] == total(sc_rand)).all() and (got[2] == got[0]).all()
    params.free()


def test_msm_skewed_buckets(gpu, O, P, srs11):
    """selector-like / sorted-lookup-like scalar vectors put most points in a few buckets:
    heavy buckets are split into tasks and folded in merge rounds"""
    from circuits_halo2_amd import ffi
    p = srs11["gl_np"][:64]
    pt = P.g1_from_bytes(p.tobytes())
    n = 1 << 13
    bases = O.fixed_base_mul(O.random_fr(21, n), O.ncpu())
    ffi.check(ffi.lib().sg_set_param(b"msm.log_seg", 2))  # L = 4: forces several merge rounds
    try:
        assert (gpu.best_multiexp(fr_np([7] * 300), np.tile(p, 300)) == point_np(P.g1_mul(pt, 2100))).all()
        ones = fr_np([1] * n)
        assert (gpu.best_multiexp(ones, bases) == O.best_multiexp(ones, bases, O.ncpu())).all()
        mixed = fr_np([(i % 3) + 1 if i % 5 else 0 for i in range(n)])
        assert (gpu.best_multiexp(mixed, bases) == O.best_multiexp(mixed, bases, O.ncpu())).all()
    finally:
        ffi.check(ffi.lib().sg_set_param(b"msm.log_seg", 0))
    ones = fr_np([1] * n)
    assert (gpu.best_multiexp(ones, bases) == O.best_multiexp(ones, bases, O.ncpu())).all()
    bytes_like = fr_np([(i * 37) % 256 for i in range(n)])  # range-check column shape
    assert (gpu.best_multiexp(bytes_like, bases) == O.best_multiexp(bytes_like, bases, O.ncpu())).all()


def test_msm_accumulate_trace_leaves_results_alone(gpu, O, capfd):
    """`msm.acc_trace` (debug: every wave of msm_accumulate records when it starts and leaves; the host tail prints the
    percentiles): same bits with it on, a line on stderr per job, nothing once it is off again"""
    from circuits_halo2_amd import ffi
    n = 1 << 12
    sc = O.random_fr(77, n)
    bases = O.fixed_base_mul(O.random_fr(78, n), O.ncpu())
    want = O.best_multiexp(sc, bases, O.ncpu())
    ffi.check(ffi.lib().sg_set_param(b"msm.acc_trace", 1))
    try:
        assert (gpu.best_multiexp(sc, bases) == want).all()
    finally:
        ffi.check(ffi.lib().sg_set_param(b"msm.acc_trace", 0))
    err = capfd.readouterr().err
    assert "[acc_trace] waves" in err and "had left by" in err
    assert (gpu.best_multiexp(sc, bases) == want).all()
    assert "[acc_trace]" not in capfd.readouterr().err


def test_msm_batch_pipeline(gpu, O):
    """sg_msm_g1_batch: independent MSMs of different sizes pipelined over two engines"""
    sizes = [1 << 12, 0, 3000, 1 << 14, 1, 5000, 1 << 13]
    pairs, want = [], []
    for i, n in enumerate(sizes):
        sc = O.random_fr(40 + i, n) if n else np.zeros(0, np.uint8)
        bs = O.fixed_base_mul(O.random_fr(60 + i, n), O.ncpu()) if n else np.zeros(0, np.uint8)
        pairs.append((sc, bs))
        want.append(O.best_multiexp(sc, bs, O.ncpu()))
    got = gpu.best_multiexp_batch(pairs)
    for g, w in zip(got, want):
        assert (g == w).all()
    dpairs = [(dev(s), dev(b)) for s, b in pairs if s.size]
    got = gpu.best_multiexp_batch(dpairs)
    for g, w in zip(got, [w for w, (s, _) in zip(want, pairs) if s.size]):
        assert (g == w).all()
    assert gpu.best_multiexp_batch([]) == []
    # equal-length MSMs are fused into one job: same bases / different scalars (advice columns),
    # different bases too, all-zero members, more members than one fused job holds
    n = 1 << 12
    bases = [O.fixed_base_mul(O.random_fr(70 + i, n), O.ncpu()) for i in range(2)]
    pairs = []
    for i in range(40):
        sc = O.random_fr(100 + i, n) if i % 7 != 3 else np.zeros(32 * n, np.uint8)
        pairs.append((sc, bases[i % 2]))
    got = gpu.best_multiexp_batch(pairs)
    for g, (sc, bs) in zip(got, pairs):
        assert (g == O.best_multiexp(sc, bs, O.ncpu())).all()


def test_fixed_base_mul(gpu, O):
    from circuits_halo2_amd.arithmetic import g1_fixed_base_mul
    sc = np.concatenate([O.random_fr(31, 200), fr_np([0, 1, 2])])
    assert (g1_fixed_base_mul(sc) == O.fixed_base_mul(sc, 4)).all()


@pytest.mark.parametrize("log_n", [17, 20])
def test_msm_full_size_known_answer(gpu, O, log_n):
    """BASELINE config 2: MSM over bases s_i*G must equal <k, s>*G (exact, independent of
    the MSM algorithm); everything device resident."""
    from circuits_halo2_amd.arithmetic import g1_fixed_base_mul, fr_to_montgomery
    from circuits_halo2_amd.utils import random_fr_canonical
    n = 1 << log_n
    k = fr_to_montgomery(dev(random_fr_canonical(0x53554D4D41, n)))
    s = fr_to_montgomery(dev(random_fr_canonical(0x7A55, n)))
    bases = g1_fixed_base_mul(s)
    got, tm = gpu.best_multiexp(k, bases, timings=True)
    dot = O.fr_dot(k.cpu().numpy(), s.cpu().numpy())
    assert (got == O.g1_mul(O.g1_generator(), dot)).all()
    assert O.g1_is_on_curve(got)
    print("msm timings", log_n, tm)
    # witness-like scalars: 99 % zero, the rest below 2^64 (advice-column shape)
    kk = k.cpu().numpy().reshape(n, 32).copy()
    rng = np.random.default_rng(5)
    keep = rng.random(n) < 0.01
    kk[~keep] = 0
    small = np.zeros((n, 32), dtype=np.uint8)
    small[:, :8] = kk[:, :8]
    small[~keep] = 0
    from circuits_halo2_amd.utils import to_montgomery_host
    idx = np.nonzero(keep)[0]
    sm = np.zeros((n, 32), dtype=np.uint8)
    sm[idx] = to_montgomery_host(small[idx].reshape(-1)).reshape(-1, 32)
    got = gpu.best_multiexp(dev(sm.reshape(-1)), bases)
    dot = O.fr_dot(sm[idx].reshape(-1).copy(), s.cpu().numpy().reshape(n, 32)[idx].reshape(-1).copy())
    assert (got == O.g1_mul(O.g1_generator(), dot)).all()


def test_params_setup_matches_bigint(gpu, O, P):
    """ParamsKZG::setup on the GPU vs the definition: g[i] = tau^i G, g_lagrange[i] = L_i(tau) G"""
    k, n = 6, 64
    tau = P.random_fr(0xBEEF, 1)[0]
    params = gpu.ParamsKZG.setup(k, fr_np([tau]))
    assert (params.g == O.fixed_base_mul(O.fr_powers(fr_np([tau]), n), 2)).all()
    w = P.omega_for(k)
    num = (pow(tau, n, P.R) - 1) * pow(n, -1, P.R) % P.R
    lag = [pow(w, i, P.R) * num * pow(tau - pow(w, i, P.R), -1, P.R) % P.R for i in range(n)]
    assert sum(lag) % P.R == 1
    assert (params.g_lagrange == O.fixed_base_mul(fr_np(lag), 2)).all()
    # the two bases commit to the same polynomial
    evals = O.random_fr(9, n)
    coeffs = gpu.EvaluationDomain(3, k).lagrange_to_coeff(evals)
    assert (params.commit_lagrange(evals) == params.commit(coeffs)).all()
    # verifier side and the RawBytes container: g2 = generator, s_g2 = tau * g2, write() / read() round trip
    assert params.g2 == P.g2_to_bytes(P.G2_GENERATOR)
    assert params.s_g2 == P.g2_to_bytes(P.g2_mul(P.G2_GENERATOR, tau))
    raw = params.write()
    assert len(raw) == 4 + 2 * 64 * n + 256
    again = gpu.ParamsKZG.read(raw)
    assert again.k == k and (again.g == params.g).all() and (again.g_lagrange == params.g_lagrange).all()
    assert again.g2 == params.g2 and again.s_g2 == params.s_g2
    params.free()


def test_params_write_reproduces_the_reference_container(gpu):
    """read() then write() gives back the reference's SRS fixture byte for byte (K1)"""
    import os
    from conftest import GOLDEN
    raw = open(os.path.join(GOLDEN, "hermez-raw-11"), "rb").read()
    assert gpu.ParamsKZG.read(raw).write() == raw


def test_k11_proof_op_shapes_on_reference_srs(gpu, O, srs11):
    """BASELINE configs[0] shape (k = 11, SRS hermez-raw-11): the MSM / NTT call shapes of one
    MstInclusion proof -- commit_lagrange of witness-like and dense columns,
    commit of coefficient-form polynomials, lagrange_to_coeff (2^11), coeff_to_extended /
    extended_to_coeff (2^14) -- each compared with the oracle."""
    k = 11
    n = 1 << k
    params = gpu.ParamsKZG(k, srs11["g_np"], srs11["gl_np"])
    dom = gpu.EvaluationDomain(6, k)
    rng = np.random.default_rng(11)
    dense = O.random_fr(0x11, n)
    bytes_col = fr_np([int(v) for v in rng.integers(0, 256, n)])          # range-check column
    sparse = fr_np([int(v) if i % 37 == 0 else 0 for i, v in enumerate(rng.integers(0, 1 << 62, n))])
    cols = [dense, bytes_col, sparse]
    got = gpu.best_multiexp_batch([(c, srs11["gl_np"]) for c in cols])          # phase-1 advice commitments
    for g, c in zip(got, cols):
        assert (g == O.best_multiexp(c, srs11["gl_np"], O.ncpu())).all()
        assert (params.commit_lagrange(c) == g).all()
    coeffs = dom.lagrange_to_coeff(dense)
    assert (coeffs == O.lagrange_to_coeff(dense, k, O.ncpu())).all()
    assert (params.commit(coeffs) == O.best_multiexp(coeffs, srs11["g_np"], O.ncpu())).all()
    assert (params.commit(coeffs) == params.commit_lagrange(dense)).all()      # same polynomial, two bases
    ext = dom.coeff_to_extended(coeffs)
    assert (ext == O.coeff_to_extended(coeffs, k, dom.extended_k, O.ncpu())).all()
    h = dom.extended_to_coeff(dom.divide_by_vanishing_poly(ext))
    want = O.extended_to_coeff(O.divide_by_vanishing_poly(ext, k, dom.extended_k), k, dom.extended_k, O.ncpu())
    assert (h == want[:h.size]).all()
    pieces = [h[32 * n * i:32 * n * (i + 1)] for i in range(5)]                # 5 quotient pieces
    got = gpu.best_multiexp_batch([(p, srs11["g_np"]) for p in pieces])
    for g, p in zip(got, pieces):
        assert (g == O.best_multiexp(np.ascontiguousarray(p), srs11["g_np"], O.ncpu())).all()
    params.free()


def test_g1_fft_reproduces_the_reference_srs_lagrange_basis(gpu, O, srs11):
    """N5 golden: the reference's own SRS file holds both bases; the inverse G1 FFT of g[]
    (what ParamsKZG::downsize / g_to_lagrange compute) must give the file's g_lagrange."""
    import ctypes as C
    from circuits_halo2_amd import ffi
    gl = np.zeros_like(srs11["gl_np"])
    ffi.check(ffi.lib().sg_g1_to_lagrange(ffi.ptr(srs11["g_np"]), C.c_uint32(11), ffi.ptr(gl)))
    assert (gl == srs11["gl_np"]).all()


def test_params_downsize(gpu, O, srs11):
    params = gpu.ParamsKZG(11, srs11["g_np"], srs11["gl_np"])
    with pytest.raises(ValueError):
        params.downsize(12)
    params.downsize(8)
    assert params.k == 8 and (params.g == srs11["g_np"][:64 * 256]).all()
    dom = gpu.EvaluationDomain(3, 8)
    for j in (0, 1, 77, 255):  # g_lagrange[j] = commit(L_j) over the truncated monomial basis
        e = [0] * 256
        e[j] = 1
        assert (params.commit(dom.lagrange_to_coeff(fr_np(e))) == params.g_lagrange[64 * j:64 * j + 64]).all()
    evals = O.random_fr(3, 256)
    assert (params.commit_lagrange(evals) == params.commit(dom.lagrange_to_coeff(evals))).all()
    params.free()


# ----------------------------------------------------------------------------- §8f-2 helpers
@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 8191, 8192, 8193, 100000, 1 << 17, (1 << 20) + 5])
def test_eval_polynomial(gpu, O, n):
    from circuits_halo2_amd.arithmetic import eval_polynomial
    c = O.random_fr(700 + n % 97, n)
    x = O.random_fr(701, 1)
    want = O.fr_eval_poly(c, x)
    assert (eval_polynomial(c, x) == want).all()
    assert (eval_polynomial(dev(c), x) == want).all()
    one = fr_np([1])
    assert (eval_polynomial(c, fr_np([0])) == c[:32]).all()
    if n <= 8193:
        tot = fr_np([sum(__import__("oracle.pyref", fromlist=["x"]).frs_from_bytes(c.tobytes())) %
                     21888242871839275222246405745257275088548364400416034343698204186575808495617])
        assert (eval_polynomial(c, one) == tot).all()


@pytest.mark.parametrize("n", [1, 7, 8, 9, 1000, 1 << 17])
def test_batch_invert_and_prefix_product(gpu, O, n):
    from circuits_halo2_amd.arithmetic import batch_invert, prefix_product, fr_mul
    a = O.random_fr(800 + n % 89, n)
    a[32 * (n // 3):32 * (n // 3) + 32] = 0          # zeros stay zero
    if n > 20:
        a[32 * 17:32 * 18] = 0
    inv = batch_invert(dev(a)).cpu().numpy()
    assert (inv == O.fr_batch_invert(a)).all()
    b = O.random_fr(801, n)
    assert (prefix_product(dev(b)).cpu().numpy() == O.fr_prefix_product(b)).all()
    assert (fr_mul(dev(a), dev(b)).cpu().numpy() == O.fr_mul_n(a, b)).all()
    # a grand product the way the permutation argument builds it: z[i+1] = z[i] * num[i] / den[i]
    num, den = O.random_fr(802, n), O.random_fr(803, n)
    ratio = fr_mul(dev(num), batch_invert(dev(den)))
    z = prefix_product(ratio).cpu().numpy()
    assert (z == O.fr_prefix_product(O.fr_mul_n(num, O.fr_batch_invert(den)))).all()


def test_sizes_beyond_the_baseline_configs(gpu, O):
    """maximum-size cases: MSM 2^22 (known answer <k,s> G) and NTT 2^24 (round trip + spot value)"""
    from circuits_halo2_amd.arithmetic import g1_fixed_base_mul, fr_to_montgomery
    from circuits_halo2_amd.utils import random_fr_canonical
    n = 1 << 22
    k = fr_to_montgomery(dev(random_fr_canonical(0xA1, n)))
    s = fr_to_montgomery(dev(random_fr_canonical(0xA2, n)))
    bases = g1_fixed_base_mul(s)
    got = gpu.best_multiexp(k, bases)
    assert (got == O.g1_mul(O.g1_generator(), O.fr_dot(k.cpu().numpy(), s.cpu().numpy()))).all()
    del bases, s
    log_n = 24
    a = fr_to_montgomery(dev(random_fr_canonical(0xA3, 1 << log_n)))
    orig = a.clone()
    dom = gpu.EvaluationDomain(2, log_n)
    gpu.best_fft(a, dom.get_omega(), log_n)
    j = 12345
    x = O.fr_powers(dom.get_omega(), j + 1)[32 * j:32 * j + 32].copy()
    from circuits_halo2_amd.arithmetic import eval_polynomial
    assert (eval_polynomial(orig, x) == a[32 * j:32 * j + 32].cpu().numpy()).all()
    dom.lagrange_to_coeff(a)
    assert bool((a == orig).all())


def test_abi_is_thread_safe(gpu, O):
    """halo2 calls best_multiexp / best_fft from inside rayon iterators: concurrent callers
    must get correct results (each call owns one of the library's lanes)"""
    import threading
    n = 1 << 12
    bases = O.fixed_base_mul(O.random_fr(950, n), O.ncpu())
    jobs = []
    for i in range(6):
        sc = O.random_fr(960 + i, n)
        a = O.random_fr(970 + i, 1 << 11)
        jobs.append((sc, O.best_multiexp(sc, bases, O.ncpu()), a, O.best_fft(a, O.omega(11), 11, 2)))
    errors = []

    def worker(sc, want_p, a, want_a):
        try:
            for _ in range(3):
                assert (gpu.best_multiexp(sc, bases) == want_p).all()
                assert (gpu.best_fft(a, O.omega(11), 11) == want_a).all()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_host_pointer_entry_points_from_many_threads(gpu, O):
    """the entry points that stage host buffers (sg_msm_g1_batch, sg_commit, sg_fr_eval_poly) keep their lane from
    staging to result: eight threads hammering them with different inputs all get the oracle's answers (the advisor's
    round-1 finding: the staging buffers used to be released between staging and compute)"""
    import threading
    from circuits_halo2_amd import arithmetic as A
    n = 1 << 11
    bases = O.fixed_base_mul(O.random_fr(1950, n), O.ncpu())
    params = gpu.ParamsKZG(11, bases, bases)
    jobs = []
    for i in range(8):
        m = n - 37 * i                              # different lengths: the staging areas are re-sized between calls
        sc = [O.random_fr(1960 + 10 * i + j, m) for j in range(3)]
        want = [O.best_multiexp(s, bases[:64 * m], 2) for s in sc]
        x = O.random_fr(1990 + i, 1)
        jobs.append((m, sc, want, x, O.fr_eval_poly(sc[0], x)))
    errors = []

    def worker(m, sc, want, x, want_eval):
        try:
            for _ in range(4):
                got = gpu.best_multiexp_batch([(s, bases[:64 * m]) for s in sc])
                assert all((g == w).all() for g, w in zip(got, want)), "batch"
                assert (params.commit(sc[1]) == want[1]).all(), "commit"
                assert (A.eval_polynomial(sc[0], x) == want_eval).all(), "eval"
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    params.free()
    assert not errors, errors


def test_sparse_hint_changes_the_schedule_not_the_commitments(gpu, O):
    """SG_BASIS_SPARSE (basis | 16): witness-like columns -- a few thousand small values, the rest zero, heavy buckets -- get
    shorter accumulation tasks when every column of the job carries the hint; with it, without it and with it on some columns
    only the commitments are the same points, for plain Lagrange columns and for difference-form ones"""
    import torch
    from circuits_halo2_amd import arithmetic as A
    k = 14
    n = 1 << k
    bases = O.fixed_base_mul(O.random_fr(3300, n), O.ncpu())
    params = gpu.ParamsKZG(k, bases, bases)
    params.precompute()
    try:
        rng = np.random.default_rng(33)
        cols = []
        for c in range(3):
            vals = np.zeros((n, 32), dtype=np.uint8)
            used = 3000 + 500 * c
            vals[:used, 0] = rng.integers(0, 256, used)             # bytes of a range check
            vals[:used:7, 1] = rng.integers(0, 4, len(vals[:used:7]))
            cols.append(A.fr_to_montgomery(torch.from_numpy(vals.reshape(-1)).cuda()))
        for flags in ([1, 1, 1], [2, 2, 1], [0, 1, 2]):
            plain = params.commit_batch_mixed(cols, flags)
            assert (params.commit_batch_mixed(cols, [f | 16 for f in flags]) == plain).all()
            assert (params.commit_batch_mixed(cols, [flags[0] | 16] + flags[1:]) == plain).all()
        want = np.stack([O.best_multiexp(c.cpu().numpy(), bases, O.ncpu()) for c in cols])
        assert (params.commit_batch_mixed(cols, [17, 17, 17]) == want).all()
        with pytest.raises(ValueError):
            params.commit_batch_mixed(cols, [3, 1, 1])
    finally:
        params.free()


@pytest.mark.parametrize("n", [(1 << 18) - 1, 1 << 18, (1 << 18) + 1, (1 << 19) + 12345])
def test_host_pointer_msm_in_two_halves(gpu, O, n):
    """sg_msm_g1 / sg_commit from host memory cut inputs of 2^18 pairs and more into chunks that run as jobs on the lane's two
    engines while a third stream carries the copies, and add the partial points on the host: the
    same point as the device path and as <k, s> G, at the threshold, around it, for an odd length, with an all-zero first
    half (its job finds nothing to do), from memory registered with sg_host_register, and under the sleeping host wait"""
    import ctypes as C
    import torch
    from circuits_halo2_amd import arithmetic as A, ffi
    from circuits_halo2_amd.utils import random_fr_canonical
    L = gpu.lib()
    sc = A.fr_to_montgomery(torch.from_numpy(random_fr_canonical(3100 + n % 97, n)).cuda())
    bs = A.fr_to_montgomery(torch.from_numpy(random_fr_canonical(3200 + n % 89, n)).cuda())
    bases = A.g1_fixed_base_mul(bs)
    one = np.frombuffer((0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb).to_bytes(32, "little"), dtype=np.uint8)
    want = np.asarray(A.g1_fixed_base_mul(A.eval_polynomial(A.fr_mul(sc, bs), one)))        # <k, s> G: bases are s_i G
    assert (gpu.best_multiexp(sc, bases) == want).all()
    hs, hb = sc.cpu().numpy().copy(), bases.cpu().numpy().copy()
    out = np.zeros(64, dtype=np.uint8)

    def host_msm(s_, b_):
        ffi.check(L.sg_msm_g1(ffi.ptr(s_), ffi.ptr(b_), C.c_size_t(s_.size // 32), ffi.ptr(out)))
        return out.copy()
    assert (host_msm(hs, hb) == want).all()
    ffi.check(L.sg_host_register(C.c_void_p(hs.ctypes.data), C.c_size_t(hs.nbytes)))
    try:
        assert L.sg_host_register(C.c_void_p(hs.ctypes.data + 64), C.c_size_t(64)) != 0        # overlaps a registered range
        assert (host_msm(hs, hb) == want).all()
        ffi.check(L.sg_set_param(b"host.wait_sleep_us", 40))
        try:
            assert (host_msm(hs, hb) == want).all()
            assert (gpu.best_multiexp(sc, bases) == want).all()
        finally:
            ffi.check(L.sg_set_param(b"host.wait_sleep_us", 0))
    finally:
        ffi.check(L.sg_host_unregister(C.c_void_p(hs.ctypes.data)))
    assert L.sg_host_unregister(C.c_void_p(hs.ctypes.data)) != 0                                # not registered any more
    # the first half all zero: that half's job has no entries; the result is the second half's
    h = n // 2
    hz = hs.copy()
    hz[:32 * h] = 0
    tail = gpu.best_multiexp(sc[32 * h:].contiguous(), bases[64 * h:].contiguous())
    assert (host_msm(hz, hb) == tail).all()
    hz[:] = 0
    assert not host_msm(hz, hb).any()                                                            # identity = 64 zero bytes
    # every chunk count, and the resident-SRS flavour (sg_commit: scalars from the host, bases in HBM, generic path)
    k_srs = (n - 1).bit_length()
    pad = (1 << k_srs) - n
    params = gpu.ParamsKZG(k_srs, np.concatenate([hb, np.zeros(64 * pad, dtype=np.uint8)]), np.concatenate([hb, np.zeros(64 * pad, dtype=np.uint8)]))
    try:
        for chunks in (1, 2, 3, 4, 8, 0):
            ffi.check(L.sg_set_param(b"msm.host_chunks", chunks))
            assert (host_msm(hs, hb) == want).all(), chunks
            ffi.check(L.sg_commit(C.c_uint64(params.handle()), C.c_int(0), ffi.ptr(hs), C.c_size_t(n), ffi.ptr(out)))
            assert (out == want).all(), chunks
    finally:
        ffi.check(L.sg_set_param(b"msm.host_chunks", 0))
        params.free()


def test_two_host_threads_do_not_serialise(gpu, O):
    """calls from different host threads take different lanes of the library (own streams, MSM engines, work space): two
    threads issuing the same MSMs concurrently must finish well before twice the time one of them needs alone (they would
    queue behind one lock otherwise), with every result still the oracle's"""
    import threading
    import time
    import torch
    n = 1 << 17
    bases = O.fixed_base_mul(O.random_fr(2950, n), O.ncpu())
    d_bases = torch.from_numpy(bases).cuda()
    scal = [O.random_fr(2960 + i, n) for i in range(2)]
    want = [O.best_multiexp(s, bases, O.ncpu()) for s in scal]
    d_scal = [torch.from_numpy(s).cuda() for s in scal]
    torch.cuda.synchronize()
    reps = 10

    def run(i, out):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for r in range(reps):
                out.append((i, gpu.best_multiexp(d_scal[i], d_bases)))

    def timed(which):
        outs = [[] for _ in which]
        threads = [threading.Thread(target=run, args=(i, o)) for i, o in zip(which, outs)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        dt = time.perf_counter() - t0
        for o in outs:
            assert len(o) == reps and all((p == want[i]).all() for i, p in o)
        return dt
    timed([0]); timed([0, 1])                                     # warm both lanes' work spaces
    alone = min(timed([0]) for _ in range(3))
    both = min(timed([0, 1]) for _ in range(3))                   # twice the work, two threads
    print(f"{reps} MSMs of 2^17: one thread {alone * 1e3:.2f} ms; two threads, {reps} each: {both * 1e3:.2f} ms")
    assert both < 1.8 * alone, (alone, both)


@pytest.mark.parametrize("k,ncols", [(4, 1), (9, 4), (11, 2), (13, 4)])
def test_permutation_grand_product(gpu, O, P, k, ncols):
    """halo2 permutation::prover::commit, one chunk (for MstInclusion: 6 columns in chunks of 4 + 2)"""
    from circuits_halo2_amd.arithmetic import permutation_product
    n = 1 << k
    vals = [O.random_fr(1100 + 10 * k + c, n) for c in range(ncols)]
    sig = [O.random_fr(1200 + 10 * k + c, n) for c in range(ncols)]
    beta, gamma = O.random_fr(1301, 1), O.random_fr(1302, 1)
    dstart = fr_np([pow(P.DELTA, 4, P.R)])          # second chunk of a chunk_len = 4 argument
    z0 = O.random_fr(1303, 1)
    want = O.permutation_product(vals, sig, beta, gamma, dstart, k, z0)
    got = permutation_product([dev(v) for v in vals], [dev(s) for s in sig], beta, gamma, dstart, k, z0)
    assert (got.cpu().numpy() == want).all()
    one = fr_np([1])
    want = O.permutation_product(vals, sig, beta, gamma, one, k)
    got = permutation_product([dev(v) for v in vals], [dev(s) for s in sig], beta, gamma, one, k)
    assert (got.cpu().numpy() == want).all()
    # a real permutation: sigma = identity labels delta^c * omega^i  =>  every fraction is 1, z == 1
    w = P.omega_for(k)
    labels = [fr_np([pow(P.DELTA, c, P.R) * pow(w, i, P.R) % P.R for i in range(n)]) for c in range(ncols)] if k <= 9 else None
    if labels:
        got = permutation_product([dev(v) for v in vals], [dev(s) for s in labels], beta, gamma, one, k)
        assert (got.cpu().numpy() == np.tile(one, n)).all()


@pytest.mark.parametrize("k,shape", [(5, ((4, 2), 1)), (11, ((4, 2), 1)), (13, ((3, 3, 1), 2)), (17, ((4, 2), 1)), (9, ((), 2)), (9, ((5,), 0))])
def test_grand_products_batched(gpu, O, P, k, shape):
    """sg_grand_products_dev: every grand product of a proof in batched launches -- the chunks of the permutation argument
    (chunk j continues from chunk j-1's value at the last usable row, on the device) and the lookups, through ONE batch
    inversion -- equals the oracle's products chained by hand, bit for bit (MstInclusion's shape is ((4, 2), 1))"""
    from circuits_halo2_amd.arithmetic import grand_products
    chunks, n_lookups = shape
    n = 1 << k
    u = n - 6
    beta, gamma = O.random_fr(2301 + k, 1), O.random_fr(2302 + k, 1)
    perm, want_z, col, z0 = [], [], 0, None
    for j, nc in enumerate(chunks):
        vals = [O.random_fr(2100 + 100 * k + col + c, n) for c in range(nc)]
        sig = [O.random_fr(2200 + 100 * k + col + c, n) for c in range(nc)]
        dstart = fr_np([pow(P.DELTA, col, P.R)])
        z = O.permutation_product(vals, sig, beta, gamma, dstart, k, z0)
        want_z.append(z)
        z0 = z[32 * u:32 * (u + 1)].copy()
        perm.append(([dev(v) for v in vals], [dev(s_) for s_ in sig]))
        col += nc
    lookups, want_l = [], []
    for l in range(n_lookups):
        a, s_, ap, sp = (O.random_fr(2400 + 10 * l + i + k, n) for i in range(4))
        want_l.append(O.lookup_product(a, s_, ap, sp, beta, gamma))
        lookups.append((dev(a), dev(s_), dev(ap), dev(sp)))
    got_z, got_l = grand_products(perm, lookups, beta, gamma, k, u)
    assert len(got_z) == len(chunks) and len(got_l) == n_lookups
    for got, want in zip(got_z + got_l, want_z + want_l):
        assert (got.cpu().numpy() == want).all()


def test_grand_products_batched_arguments(gpu, O):
    from circuits_halo2_amd.arithmetic import grand_products
    k = 6
    col = lambda seed: dev(O.random_fr(seed, 1 << k))
    beta, gamma = O.random_fr(1, 1), O.random_fr(2, 1)
    assert grand_products([], [], beta, gamma, k, 10) == ([], [])
    with pytest.raises(gpu.SummaGpuError):       # usable rows beyond the domain
        grand_products([([col(3)], [col(4)])], [], beta, gamma, k, 1 << k)
    with pytest.raises(gpu.SummaGpuError):       # more than eight products
        grand_products([([col(3)], [col(4)])] * 9, [], beta, gamma, k, 10)
    with pytest.raises(gpu.SummaGpuError):       # more than eight columns in a chunk
        grand_products([([col(3)] * 9, [col(4)] * 9)], [], beta, gamma, k, 10)


@pytest.mark.parametrize("n", [16, 2048, 1 << 13, 5000])
def test_lookup_grand_product(gpu, O, n):
    from circuits_halo2_amd.arithmetic import lookup_product
    a, s, ap, sp = (O.random_fr(1400 + i, n) for i in range(4))
    beta, gamma = O.random_fr(1411, 1), O.random_fr(1412, 1)
    want = O.lookup_product(a, s, ap, sp, beta, gamma)
    got = lookup_product(dev(a), dev(s), dev(ap), dev(sp), beta, gamma)
    assert (got.cpu().numpy() == want).all()
    # permuted columns that are permutations of the inputs => the product telescopes back to 1
    perm = np.random.default_rng(3).permutation(n)
    ap2 = a.reshape(n, 32)[perm].reshape(-1).copy()
    sp2 = s.reshape(n, 32)[perm].reshape(-1).copy()
    z = lookup_product(dev(a), dev(s), dev(ap2), dev(sp2), beta, gamma).cpu().numpy()
    last = O.fr_mul(z[-32:].copy(), O.fr_mul(O.fr_mul(O.fr_add(a[-32:].copy(), beta), O.fr_add(s[-32:].copy(), gamma)),
                                             O.fr_inv(O.fr_mul(O.fr_add(ap2[-32:].copy(), beta), O.fr_add(sp2[-32:].copy(), gamma)))))
    assert (last == fr_np([1])).all()


@pytest.mark.parametrize("k,j", [(4, 6), (9, 6), (13, 6), (15, 6), (16, 3), (17, 6)])
def test_coeff_to_extended_batch(gpu, O, k, j):
    """several columns per launch (extended domain <= 2^18) or one after the other: same bits as the oracle"""
    dom = gpu.EvaluationDomain(j, k)
    cols = [O.random_fr(2800 + i, 1 << k) for i in range(5)]
    outs = dom.coeff_to_extended_batch([dev(c) for c in cols])
    for c, o in zip(cols, outs):
        assert (o.cpu().numpy() == O.coeff_to_extended(c, k, dom.extended_k, O.ncpu())).all()
    assert dom.coeff_to_extended_batch([]) == []


@pytest.mark.parametrize("log_n", [9, 13, 17])
def test_ntt_batch(gpu, O, log_n):
    from circuits_halo2_amd.arithmetic import best_fft_batch
    vecs = [O.random_fr(1500 + i, 1 << log_n) for i in range(5)]
    d = [dev(v) for v in vecs]
    best_fft_batch(d, O.omega(log_n), log_n)
    for got, v in zip(d, vecs):
        assert (got.cpu().numpy() == O.best_fft(v, O.omega(log_n), log_n, O.ncpu())).all()
    best_fft_batch(d, O.omega_inv(log_n), log_n, divisor=O.n_inv(log_n))
    for got, v in zip(d, vecs):
        assert (got.cpu().numpy() == v).all()


def test_independent_ops_on_several_streams(gpu, O):
    """the work space of the NTT and of the scan-type helpers is per stream: in-place multi-pass transforms, grand
    products and Kate divisions issued on three streams at once give the same bits as the oracle"""
    import torch
    from circuits_halo2_amd.arithmetic import kate_division, lookup_product, permutation_product
    k = 14
    n = 1 << k
    dom = gpu.EvaluationDomain(6, k)
    streams = [torch.cuda.Stream() for _ in range(3)]
    cols = [O.random_fr(2700 + i, n) for i in range(9)]
    beta, gamma, one, b = O.random_fr(2710, 1), O.random_fr(2711, 1), fr_np([1]), O.random_fr(2712, 1)
    d = [dev(c) for c in cols]
    torch.cuda.synchronize()
    res = {}
    for rep in range(3):                      # several rounds: buffers are reused while other streams still run
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                a = d[3 * si].clone()
                res[("coeff", si)] = dom.lagrange_to_coeff(a)                       # 3-pass, in place, scratch
                res[("ext", si)] = dom.coeff_to_extended(res[("coeff", si)])
                res[("z", si)] = permutation_product([d[3 * si], d[3 * si + 1]], [d[3 * si + 2], d[3 * si]], beta, gamma, one, k)
                res[("zl", si)] = lookup_product(d[3 * si], d[3 * si + 1], d[3 * si + 2], d[3 * si], beta, gamma)
                res[("q", si)] = kate_division(d[3 * si + 1], b)
    torch.cuda.synchronize()
    for si in range(3):
        c0, c1, c2 = cols[3 * si], cols[3 * si + 1], cols[3 * si + 2]
        want_c = O.lagrange_to_coeff(c0, k)
        assert (res[("coeff", si)].cpu().numpy() == want_c).all()
        assert (res[("ext", si)].cpu().numpy() == O.coeff_to_extended(want_c, k, dom.extended_k)).all()
        assert (res[("z", si)].cpu().numpy() == O.permutation_product([c0, c1], [c2, c0], beta, gamma, one, k)).all()
        assert (res[("zl", si)].cpu().numpy() == O.lookup_product(c0, c1, c2, c0, beta, gamma)).all()
        assert (res[("q", si)].cpu().numpy() == O.fr_kate_division(c1, b)[0]).all()


@pytest.mark.parametrize("overlap", [False, True])
def test_proof_flow_schedule_runs(gpu, overlap):
    """tools/proof_flow.py (what bench.py reports as proof_flow_k17) at a small size: every phase executes, on one
    stream and with the side-stream schedule"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from proof_flow import run_flow
    t = run_flow(9, n_gates=3, reps=1, overlap=overlap)
    assert set(t) >= {"1_advice_commit", "3_grand_products_commit", "4b_evaluate_h", "6_multiopen", "total"}
    assert t["total"] > 0


def test_cpp_proof_flow_driver_runs(gpu):
    """tools/proof_flow.cpp: the op schedule driven from C++ over the C ABI (built by __graft_entry__.build())"""
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "proof_flow_cpp")
    if not os.path.exists(exe):
        pytest.skip("tools/proof_flow_cpp not built")
    out = subprocess.run([exe, "9", "3", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    t = json.loads(out.stdout.strip().splitlines()[-1])
    assert t["driver"] == "c++" and t["total"] > 0 and "6_multiopen" in t


def test_cpp_host_mirror_parity(gpu):
    """include/summa_gpu.hpp (best_multiexp, best_fft, EvaluationDomain, ParamsKZG in C++ over the C ABI) against
    the oracle: tests/cpp/parity_main.cpp, built by __graft_entry__.build()"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "parity_main")
    if not os.path.exists(exe):
        pytest.skip("tests/cpp/parity_main not built")
    out = subprocess.run([exe, os.path.join(root, "tests", "golden", "hermez-raw-11")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout


def test_quad_cooperative_add_selftest(gpu):
    """xyzz29_add_quad (4 lanes per point addition, DPP exchanges) against the one-lane formulas on
    1024 operand pairs including P + P, P + (-P) and identity operands (tools/test_quad.hip, built by
    __graft_entry__.build())"""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "test_quad")
    if not os.path.exists(exe):
        pytest.skip("tools/test_quad not built")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatching lanes: 0 of" in out.stdout


@pytest.mark.parametrize("nc", [1, 2, 3, 4])
def test_ahead_of_time_gate_program_equals_the_interpreter(gpu, capfd, nc):
    """the reference circuit's gate programs are also compiled ahead of time (gates_mst_programs.inc: every instruction a
    template instantiation, values in registers); sg_quotient_gates* picks that kernel when the program it is given lowers
    to exactly the table.  Same rows as the interpreter (SG_GATES_GENERIC=1), on the extended domain and on cosets, and the
    ahead-of-time kernel is the one that ran"""
    import os
    import torch
    from circuits_halo2_amd import arithmetic as A, mst_inclusion as M
    from circuits_halo2_amd.utils import random_fr_canonical
    k, ext_k, d = 7, 10, 5
    graph = M.gate_graph(nc)
    chal = M.gate_challenges(0x1234567 + nc, nc)
    b = np.frombuffer(bytes(range(1, 33)), dtype=np.uint8).copy(); b[31] = 0
    for rows, kw in ((1 << ext_k, None), (d << k, d)):
        col = lambda s: A.fr_to_montgomery(torch.from_numpy(random_fr_canonical(s, rows)).cuda())
        fixed = [col(100 + i) for i in range(M.NUM_FIXED)]
        advice = [col(200 + i) for i in range(M.NUM_ADVICE)]
        start = col(300)
        outs = []
        for generic in (False, True):
            if generic:
                os.environ["SG_GATES_GENERIC"] = "1"
            os.environ["SG_GATES_DEBUG"] = "1"
            try:
                v = start.clone()
                if kw is None:
                    A.quotient_gates(v, graph, fixed, advice, [], chal, b, b, b, b, k, ext_k)
                else:
                    A.quotient_gates_cosets(v, graph, fixed, advice, [], chal, b, b, b, b, k, kw)
                torch.cuda.synchronize()
            finally:
                os.environ.pop("SG_GATES_GENERIC", None)
                os.environ.pop("SG_GATES_DEBUG", None)
            err = capfd.readouterr().err
            assert ("ahead-of-time program MstGatesNc%d" % nc in err) == (not generic), err
            outs.append(v)
        assert (outs[0] == outs[1]).all()
        assert not (outs[0] == start).all()


def test_field_products_device_selftest(gpu):
    """the device spelling of the field products (column chains of v_mad_u64_u32 in inline asm: f29_mul, f29_sqr,
    f29_mul2, f29_mul_add, f29_dot<2..5>) against the plain C++ definition on the host, limb for limb: random
    operands, operands at the top of their lazy bounds, edge values, both fields (tools/test_f29_device.hip, built by
    __graft_entry__.build())"""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "test_f29_device")
    if not os.path.exists(exe):
        pytest.skip("tools/test_f29_device not built")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all products agree" in out.stdout


# ----------------------------------------------------------------------------- fixed-base commits (precomputed SRS)
def test_fixed_base_goldens(gpu, O, P, srs11, kat):
    """the reference-derived known answers again, through the precomputed-window path"""
    params = gpu.ParamsKZG(11, srs11["g_np"], srs11["gl_np"])
    params.precompute()
    want = point_np((int(kat["fixed_comms"][4][0], 16), int(kat["fixed_comms"][4][1], 16)))
    col = fr_np(list(range(256)) + [0] * (2048 - 256))
    assert (params.commit_lagrange(col) == want).all()
    assert (params.commit_lagrange(dev(col)) == want).all()
    assert (params.commit_lagrange(fr_np(range(256))) == want).all()          # shorter than the table
    one = fr_np([1])
    assert (params.commit_lagrange(O.fr_powers(one, 2048)) == srs11["g_np"][:64]).all()
    wk = fr_np([pow(P.omega_for(11), 7, P.R)])
    assert (params.commit_lagrange(O.fr_powers(wk, 2048)) == srs11["g_np"][64 * 7:64 * 8]).all()
    dom = gpu.EvaluationDomain(6, 11)
    for j in (0, 5, 2047):
        e = [0] * 2048
        e[j] = 1
        assert (params.commit(dom.lagrange_to_coeff(fr_np(e))) == srs11["gl_np"][64 * j:64 * j + 64]).all()
    params.free()


@pytest.mark.parametrize("window_bits", [0, 4, 7, 11, 16])
def test_fixed_base_vs_oracle_and_generic(gpu, O, srs11, window_bits):
    """every window width: fixed-base == generic GPU path == oracle, full / ragged / edge-case scalars"""
    params = gpu.ParamsKZG(11, srs11["g_np"], srs11["gl_np"])
    params.precompute(window_bits=window_bits)
    r = P_R()
    cases = [O.random_fr(1900 + window_bits, 2048), O.random_fr(1901, 1), O.random_fr(1902, 1000),
             fr_np([0] * 2048), fr_np([1] * 2048), fr_np([r - 1] * 2048), fr_np([r - 1, 0, 1, 2, (r - 1) // 2, (r + 1) // 2] * 300),
             fr_np([(1 << 253) + 12345] * 64 + [0] * 64)]
    for sc in cases:
        n = sc.size // 32
        for basis, name in ((0, "g_np"), (1, "gl_np")):
            want = O.best_multiexp(sc, srs11[name][:64 * n], O.ncpu())
            got = params.commit(sc) if basis == 0 else params.commit_lagrange(sc)
            assert (got == want).all()
            assert (gpu.best_multiexp(sc, srs11[name][:64 * n]) == want).all()
    assert (params.commit(np.zeros(0, dtype=np.uint8)) == 0).all()
    params.free()


def test_fixed_base_batch(gpu, O, srs11):
    params = gpu.ParamsKZG(11, srs11["g_np"], srs11["gl_np"])
    cols = [O.random_fr(1950 + i, 2048) for i in range(7)] + [fr_np([0] * 2048), fr_np([5] * 2048)]
    want = np.stack([O.best_multiexp(c, srs11["gl_np"], O.ncpu()) for c in cols])
    d = [dev(c) for c in cols]
    assert (params.commit_batch(d, lagrange=True) == want).all()              # no table yet: fused generic jobs
    params.precompute(1)
    assert (params.commit_batch(d, lagrange=True) == want).all()              # fixed-base fused jobs
    assert (params.commit_batch(d[:1], lagrange=True) == want[:1]).all()
    short = [t[:32 * 700] for t in d[:3]]
    want_s = np.stack([O.best_multiexp(c[:32 * 700], srs11["g_np"][:64 * 700], O.ncpu()) for c in cols[:3]])
    params.precompute(0, window_bits=9)
    assert (params.commit_batch(short) == want_s).all()
    assert params.commit_batch([]).shape == (0, 64)
    # one basis per polynomial: generic (one table only, different plans) and fixed-base (both tables, one plan)
    flags = [True, False, True, True, False]
    want_m = np.stack([O.best_multiexp(c, srs11["gl_np" if f else "g_np"], O.ncpu()) for c, f in zip(cols[:5], flags)])
    assert (params.commit_batch_mixed(d[:5], flags) == want_m).all()
    params.precompute()
    assert (params.commit_batch_mixed(d[:5], flags) == want_m).all()
    assert (params.commit_batch_mixed(d[1:2], [False]) == want_m[1:2]).all()
    params.free()


@pytest.mark.parametrize("window_bits", [0, 5, 16])
def test_difference_form_commitments(gpu, O, srs11, window_bits):
    """basis 2 of sg_commit*: sum_i s_i L_i = sum_i (s_i - s_{i+1}) Q_i with Q the prefix sums of g_lagrange (s_n = 0).  The
    commitment must be the one of commit_lagrange for every kind of column -- random (every Q_i enters), piecewise constant
    (the case it is for: grand products over unused rows, sorted lookup columns), constant, edge values -- single, batched
    and mixed with the other bases in one fused job; without the table, and for short columns, it is plain commit_lagrange"""
    params = gpu.ParamsKZG(11, srs11["g_np"], srs11["gl_np"])
    n, r = 2048, P_R()
    rng = np.random.default_rng(77 + window_bits)
    steps = O.random_fr(2900, 40).reshape(40, 32)
    runs = np.sort(rng.integers(0, n, 39))
    piecewise = np.concatenate([np.tile(steps[j], (hi - lo, 1)) for j, (lo, hi) in enumerate(zip([0] + list(runs), list(runs) + [n]))]).reshape(-1)
    grand = piecewise.copy()
    grand[32 * (n - 6):] = O.random_fr(2901, 6)                        # blinding rows at the end, as in a z column
    cols = [O.random_fr(2902 + window_bits, n), piecewise, grand, fr_np([7] * n), fr_np([0] * n), fr_np([r - 1] * n),
            fr_np([0] * (n - 1) + [1]), fr_np([1] + [0] * (n - 1)), fr_np([r - 1, 0] * (n // 2)), fr_np(sorted(int(v) for v in rng.integers(0, 256, n)))]
    want = np.stack([O.best_multiexp(c, srs11["gl_np"], O.ncpu()) for c in cols])
    d = [dev(c) for c in cols]
    assert (params.commit_batch(d, lagrange=True, diff=True) == want).all()      # no tables at all: generic path
    params.precompute(1, window_bits=window_bits)
    assert (params.commit_batch(d, lagrange=True, diff=True) == want).all()      # no prefix table: fixed-base Lagrange
    params.precompute(2, window_bits=window_bits)
    assert (params.commit_batch(d, lagrange=True, diff=True) == want).all()      # difference form
    assert (params.commit_batch(d, lagrange=True) == want).all()
    for j in (0, 1, 3, 6):
        assert (params.commit_batch(d[j:j + 1], lagrange=True, diff=True) == want[j:j + 1]).all()
    # short columns cannot telescope (s_n = 0 is needed): served as commit_lagrange
    short = [t[:32 * 700] for t in d[:3]]
    want_s = np.stack([O.best_multiexp(c[:32 * 700], srs11["gl_np"][:64 * 700], O.ncpu()) for c in cols[:3]])
    assert (params.commit_batch(short, lagrange=True, diff=True) == want_s).all()
    # mixed fused jobs: coefficients, Lagrange and difference-form columns side by side (the grand-product phase)
    flags = [2, 0, 2, 1, 2, 0, 2]
    want_m = np.stack([O.best_multiexp(c, srs11["g_np" if f == 0 else "gl_np"], O.ncpu()) for c, f in zip(cols[:7], flags)])
    assert (params.commit_batch_mixed(d[:7], flags) == want_m).all()            # tables 1 and 2 only: generic mixed path
    params.precompute(0, window_bits=window_bits)
    assert (params.commit_batch_mixed(d[:7], flags) == want_m).all()            # all three tables, one plan
    params.precompute(2, window_bits=9 if window_bits != 9 else 8)
    assert (params.commit_batch_mixed(d[:7], flags) == want_m).all()            # prefix table on another plan: falls back
    assert (params.commit_batch(d[:3], lagrange=True, diff=True) == want[:3]).all()
    with pytest.raises(ValueError):
        params.commit_batch_mixed(d[:1], [3])
    params.free()


def test_difference_form_large_known_answer(gpu, O):
    """k = 17 with a synthetic SRS: a z-like column (a few thousand distinct rows, then one value for the other 120 000,
    blinding rows at the end) commits to the same point in difference form as in the plain fixed-base form"""
    k = 17
    n = 1 << k
    params = gpu.ParamsKZG.setup(k, O.random_fr(2950, 1))
    col = O.random_fr(2951, n).copy().reshape(n, 32)
    col[6000:n - 6] = col[5999]
    col = col.reshape(-1)
    cols = [dev(col), dev(O.random_fr(2952, n)), dev(np.tile(O.random_fr(2953, 1), n))]
    params.precompute(1)
    want = params.commit_batch(cols, lagrange=True)
    params.precompute(2)
    assert (params.commit_batch(cols, lagrange=True, diff=True) == want).all()
    params.precompute(0)
    assert (params.commit_batch_mixed(cols + cols[:1], [2, 2, 2, 1]) == np.concatenate([want, want[:1]])).all()
    params.free()


def test_fixed_base_large_known_answer(gpu, O, P):
    """2^17 points of a synthetic SRS g[i] = tau^i G: sum s_i tau^i is known in the exponent"""
    k = 17
    n = 1 << k
    tau = O.random_fr(1970, 1)
    params = gpu.ParamsKZG.setup(k, tau)
    sc = O.random_fr(1971, n)
    e = O.fr_eval_poly(sc, tau)
    want = O.fixed_base_mul(e, 1)
    assert (params.commit(sc) == want).all()
    params.precompute(0)
    assert (params.commit(sc) == want).all()
    assert (params.commit(dev(sc)) == want).all()
    got = params.commit_batch([dev(sc)] * 5 + [dev(O.random_fr(1972, n))])
    assert (got[:5] == want).all()
    assert (got[5] == O.fixed_base_mul(O.fr_eval_poly(O.random_fr(1972, n), tau), 1)).all()
    # skewed scalars: one value everywhere -> every digit of a window lands in one bucket
    same = np.tile(O.random_fr(1973, 1), n)
    assert (params.commit(dev(same)) == O.fixed_base_mul(O.fr_eval_poly(same, tau), 1)).all()
    params.free()


# ----------------------------------------------------------------------------- §8f-3 pieces: multi-open helpers
@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 2047, 2048, 2049, 5000, 1 << 17, (1 << 20) + 3])
def test_kate_division(gpu, O, n):
    """halo2 kate_division: bit-exact vs the oracle recurrence; a(z) == q(z) (z - b) + a(b) at a random z"""
    from circuits_halo2_amd.arithmetic import kate_division, eval_polynomial
    a, b, z = O.random_fr(2400 + n % 97, n), O.random_fr(2401, 1), O.random_fr(2402, 1)
    want_q, want_rem = O.fr_kate_division(a, b)
    q, rem = kate_division(dev(a), b, with_remainder=True)
    assert (rem == want_rem).all() and (rem == O.fr_eval_poly(a, b)).all()
    assert (q.cpu().numpy() == want_q).all()
    if n > 1:
        lhs = O.fr_eval_poly(a, z)
        rhs = O.fr_add(O.fr_mul(eval_polynomial(q, z), O.fr_sub(z, b)), rem)
        assert (lhs == rhs).all()
    zero = fr_np([0])                                    # division by X: a shift
    q0 = kate_division(dev(a), zero)
    assert (q0.cpu().numpy() == a[32:]).all()


@pytest.mark.parametrize("n,m", [(1, 1), (255, 3), (256, 2), (100003, 16)])
def test_count_noncanonical(gpu, O, n, m):
    """sg_fr_count_noncanonical_dev against integer comparison with r: canonical data counts 0; planted words r, r + 1,
    2^256 - 1 and words that differ from r only in one 32-bit word are counted exactly"""
    import torch
    from circuits_halo2_amd.arithmetic import count_noncanonical
    from oracle import pyref as PR
    cols = [O.random_fr(2800 + j, n).copy() for j in range(m)]
    assert int(count_noncanonical([dev(c) for c in cols]).item()) == 0
    rng = np.random.default_rng(n + m)
    planted = [PR.R, PR.R + 1, (1 << 256) - 1, PR.R - 1, PR.R + (1 << 32), PR.R - (1 << 32), PR.R + (1 << 224), PR.R - (1 << 224), 0]
    want = 0
    seen = set()
    for v in planted:
        j, i = int(rng.integers(m)), int(rng.integers(n))
        if (j, i) in seen:
            continue
        seen.add((j, i))
        cols[j][32 * i:32 * i + 32] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
        want += v >= PR.R
    assert int(count_noncanonical([dev(c) for c in cols]).item()) == want
    assert int(count_noncanonical([]).item()) == 0


@pytest.mark.parametrize("n,m", [(1, 2), (7, 3), (2048, 4), (2049, 16), (1 << 13, 11), (1 << 17, 11), (300001, 5)])
def test_kate_division_batch(gpu, O, n, m):
    """sg_fr_kate_division_batch_dev: every quotient bit-exact vs the oracle's recurrence (the same polynomial may be divided
    by several points); and the multi-open's use of it: for f vanishing on a set S, sum_j c_j f / (X - p_j) with the Lagrange
    denominators c_j equals the chained division f / prod_j (X - p_j)"""
    from circuits_halo2_amd.arithmetic import kate_division, kate_division_batch, lincomb
    polys = [O.random_fr(2700 + j % 3, n) for j in range(m)]
    d_polys = [dev(p) for p in polys]
    for j in range(3, m):
        d_polys[j] = d_polys[j % 3]                      # repeated inputs, as in the multi-open
        polys[j] = polys[j % 3]
    pts = O.random_fr(2750 + m, m)
    got = kate_division_batch(d_polys, pts)
    for j in range(m):
        want_q, _ = O.fr_kate_division(polys[j], pts[32 * j:32 * j + 32].copy())
        q = got[j].cpu().numpy()
        assert (q[:32 * (n - 1)] == want_q).all() and not q[32 * (n - 1):].any(), j
    if n >= 8:
        from oracle import pyref as PR
        # f = g * (X - p0)(X - p1)(X - p2): vanishes on {p0, p1, p2}
        g = O.random_fr(2790, n - 3)
        p = [PR.fr_from_bytes(bytes(pts[32 * j:32 * j + 32])) for j in range(3)]
        f = np.concatenate([g, np.zeros(96, dtype=np.uint8)])
        for pj in p:                                      # multiply by (X - pj) with integers (small n) or on the oracle
            shifted = np.concatenate([np.zeros(32, dtype=np.uint8), f[:-32]])
            scaled = np.concatenate([O.fr_mul(f[32 * i:32 * i + 32].copy(), fr_np([PR.R - pj])) for i in range(n)]) if n <= 4096 else None
            if scaled is None:
                break
            f = np.concatenate([O.fr_add(shifted[32 * i:32 * i + 32].copy(), scaled[32 * i:32 * i + 32].copy()) for i in range(n)])
        else:
            df = dev(f)
            qs = kate_division_batch([df, df, df], pts[:96].copy())
            c = [pow((p[j] - p[(j + 1) % 3]) * (p[j] - p[(j + 2) % 3]) % PR.R, -1, PR.R) for j in range(3)]
            combined = lincomb(qs, fr_np(c)).cpu().numpy()
            assert (combined[:32 * (n - 3)] == g).all() and not combined[32 * (n - 3):].any()
            chained = df
            for j in range(3):
                import torch
                chained = torch.cat([kate_division(chained, pts[32 * j:32 * j + 32].copy()), torch.zeros(32, dtype=torch.uint8, device="cuda")])
            assert (chained.cpu().numpy() == combined).all()
    with pytest.raises(ValueError):
        kate_division_batch(d_polys[:1] * 17, np.tile(pts[:32], 17))


@pytest.mark.parametrize("n,m", [(1, 3), (100, 1), (8192, 5), (8193, 2), (1 << 17, 35), (300000, 41)])
def test_eval_polynomial_batch(gpu, O, n, m):
    from circuits_halo2_amd.arithmetic import eval_polynomial_batch
    polys = [O.random_fr(2600 + j % 7, n) for j in range(m)]
    pts = O.random_fr(2650 + m, m)
    got = eval_polynomial_batch([dev(p) for p in polys], pts)
    for j in range(m):
        assert (got[j] == O.fr_eval_poly(polys[j], pts[32 * j:32 * j + 32].copy())).all()


def test_lincomb(gpu, O):
    from circuits_halo2_amd.arithmetic import lincomb
    for m, n in [(1, 5), (2, 1000), (5, 4096), (31, 3000), (32, 513)]:
        polys = [O.random_fr(2500 + j, n) for j in range(m)]
        coeffs = O.random_fr(2550 + m, m)
        got = lincomb([dev(p) for p in polys], coeffs)
        assert (got.cpu().numpy() == O.fr_lincomb(polys, coeffs)).all()
    from circuits_halo2_amd.ffi import SummaGpuError
    with pytest.raises(SummaGpuError):
        lincomb([dev(O.random_fr(1, 4))] * 33, O.random_fr(2, 33))


# ----------------------------------------------------------------------------- §8f-1: custom gates (GraphEvaluator)
def _random_graph(rng, n_fixed, n_advice, n_instance, n_chal, n_calc, O):
    """a random, valid GraphEvaluator program that exercises every calculation and value source"""
    from circuits_halo2_amd import arithmetic as A
    g = A.GraphEvaluator()
    consts = [g.add_constant(O.random_fr(int(rng.integers(1 << 30)), 1)) for _ in range(3)]
    consts.append(g.add_constant(fr_np([0])))
    consts.append(g.add_constant(fr_np([1])))
    rots = [0, 1, -1, 2, -5]

    def src(upto):
        kind = int(rng.integers(0, 8))
        if kind == 0:
            return consts[int(rng.integers(len(consts)))]
        if kind in (1, 2) and upto:
            return (A.INTERMEDIATE, int(rng.integers(max(0, upto - 6), upto)), 0)   # mostly recent values
        if kind == 3 and n_fixed:
            return g.query(A.FIXED, int(rng.integers(n_fixed)), rots[int(rng.integers(len(rots)))])
        if kind == 4 and n_instance:
            return g.query(A.INSTANCE, int(rng.integers(n_instance)), rots[int(rng.integers(2))])
        if kind == 5 and n_chal:
            return (A.CHALLENGE, int(rng.integers(n_chal)), 0)
        if kind == 6:
            return (int(rng.choice([A.BETA, A.GAMMA, A.THETA, A.Y, A.PREVIOUS_VALUE])), 0, 0)
        return g.query(A.ADVICE, int(rng.integers(n_advice)), rots[int(rng.integers(len(rots)))])

    while len(g.calculations) < n_calc:
        q = len(g.calculations)
        op = int(rng.choice([A.ADD, A.ADD, A.SUB, A.SUB, A.MUL, A.MUL, A.SQUARE, A.DOUBLE, A.NEGATE, A.HORNER, A.STORE]))
        if op == A.HORNER:
            g.add_calculation(op, src(q), src(q), [src(q) for _ in range(int(rng.integers(0, 5)))])
        elif op in (A.SQUARE, A.DOUBLE, A.NEGATE, A.STORE):
            g.add_calculation(op, src(q))
        else:
            g.add_calculation(op, src(q), src(q))
    return g


@pytest.mark.parametrize("seed,k,ext_k,n_calc", [(1, 4, 6, 8), (2, 5, 7, 40), (3, 9, 12, 120), (4, 10, 13, 300),
                                                  (5, 6, 6, 60), (6, 7, 10, 200)])
def test_quotient_gates_random_programs(gpu, O, seed, k, ext_k, n_calc):
    """bit-exact against the oracle's per-row interpreter on random programs: lazy-reduction bound tracking,
    slot allocation, rotations, Horner, aliases"""
    from circuits_halo2_amd.arithmetic import quotient_gates
    rng = np.random.default_rng(seed)
    ne = 1 << ext_k
    nf, na, ni, nc = 3, 4, 1, 2
    g = _random_graph(rng, nf, na, ni, nc, n_calc, O)
    cols = [[O.random_fr(2000 + 100 * seed + 10 * j + i, ne) for i in range(n)] for j, n in enumerate((nf, na, ni))]
    chal = O.random_fr(2090 + seed, nc)
    beta, gamma, theta, y = (O.random_fr(2095 + i, 1) for i in range(4))
    start = O.random_fr(2099, ne)
    want = O.quotient_gates(start, g.as_dict(), *cols, chal, beta, gamma, theta, y, k, ext_k)
    got = quotient_gates(dev(start), g, *[[dev(c) for c in grp] for grp in cols], chal, beta, gamma, theta, y, k, ext_k)
    assert (got.cpu().numpy() == want).all()


def test_quotient_gates_bounds_and_slots(gpu, O):
    """long lazy chains (sums of 200 terms, alternating subtractions / negations / doublings) and many live
    values (40 loads consumed in reverse order: the 64-rows-per-workgroup layout)"""
    from circuits_halo2_amd import arithmetic as A
    k, ext_k = 6, 8
    ne = 1 << ext_k
    adv = [O.random_fr(2200 + i, ne) for i in range(40)]
    beta, gamma, theta, y = (O.random_fr(2295 + i, 1) for i in range(4))
    g = A.GraphEvaluator()
    acc = g.query(A.ADVICE, 0, 0)
    for i in range(200):
        term = g.query(A.ADVICE, i % 40, (i % 3) - 1)
        op = [A.ADD, A.SUB, A.ADD, A.SUB][i % 4]
        acc = g.add_calculation(op, acc, term)
        if i % 17 == 0:
            acc = g.add_calculation(A.DOUBLE, acc)
        if i % 23 == 0:
            acc = g.add_calculation(A.NEGATE, acc)
    g.add_calculation(A.MUL, acc, acc)
    args = ([], adv, [], np.zeros(0, dtype=np.uint8), beta, gamma, theta, y, k, ext_k)
    start = O.random_fr(2299, ne)
    want = O.quotient_gates(start, g.as_dict(), *args)
    got = A.quotient_gates(dev(start), g, [], [dev(c) for c in adv], [], *args[3:])
    assert (got.cpu().numpy() == want).all()
    # 40 values loaded first, consumed last-in-first-out
    g = A.GraphEvaluator()
    loads = [g.add_calculation(A.SQUARE, g.query(A.ADVICE, i, 0)) for i in range(40)]
    acc = loads[-1]
    for v in reversed(loads[:-1]):
        acc = g.add_calculation(A.MUL, acc, v)
    want = O.quotient_gates(start, g.as_dict(), *args)
    got = A.quotient_gates(dev(start), g, [], [dev(c) for c in adv], [], *args[3:])
    assert (got.cpu().numpy() == want).all()


def test_quotient_gates_errors(gpu, O):
    from circuits_halo2_amd import arithmetic as A
    from circuits_halo2_amd.ffi import SummaGpuError
    k, ext_k = 4, 6
    ne = 1 << ext_k
    col = dev(O.random_fr(1, ne))
    b = O.random_fr(2, 1)
    run = lambda g, adv: A.quotient_gates(dev(O.random_fr(3, ne)), g, [], adv, [], np.zeros(0, dtype=np.uint8), b, b, b, b, k, ext_k)
    g = A.GraphEvaluator()
    g.add_calculation(A.ADD, g.query(A.ADVICE, 1, 0), g.query(A.ADVICE, 0, 0))       # column 1 of 1
    with pytest.raises(SummaGpuError):
        run(g, [col])
    g = A.GraphEvaluator()
    g.add_calculation(A.ADD, (A.INTERMEDIATE, 0, 0), g.query(A.ADVICE, 0, 0))        # use before definition
    with pytest.raises(SummaGpuError):
        run(g, [col])
    with pytest.raises(SummaGpuError):
        run(A.GraphEvaluator(), [col])                                               # empty program
    g = A.GraphEvaluator()                                     # 70 computed values, each used twice, far apart
    xs = [g.add_calculation(A.SQUARE, g.query(A.ADVICE, 0, i - 35)) for i in range(70)]
    acc = xs[0]
    for v in xs[1:]:
        acc = g.add_calculation(A.MUL, acc, v)
    for v in reversed(xs):
        acc = g.add_calculation(A.ADD, g.add_calculation(A.MUL, acc, v), v)
    with pytest.raises(SummaGpuError):
        run(g, [col])
    with pytest.raises(ValueError):
        run(g, [col[:64]])


# ----------------------------------------------------------------------------- §8f-1: quotient numerator
def _to_extended_dev(cols, k, ext_k):
    """Lagrange columns (numpy) -> extended-coset evaluations on the device, through the product path"""
    from circuits_halo2_amd.domain import EvaluationDomain
    dom = EvaluationDomain(1 + (1 << (ext_k - k)), k)
    assert dom.extended_k == ext_k
    return [dom.coeff_to_extended(dom.lagrange_to_coeff(dev(c))) for c in cols], dom


@pytest.mark.parametrize("k,ext_k,ncols,chunk_len", [(4, 6, 3, 2), (9, 11, 6, 4), (10, 13, 6, 4), (12, 14, 16, 2),
                                                     (11, 13, 11, 11)])
def test_quotient_permutation_random_parity(gpu, O, k, ext_k, ncols, chunk_len):
    """bit-exact against the oracle on arbitrary (non-satisfying) data, non-zero running value"""
    from circuits_halo2_amd.arithmetic import quotient_permutation
    ne = 1 << ext_k
    nsets = -(-ncols // chunk_len)
    r = lambda seed: O.random_fr(seed, ne)
    zs = [r(1600 + i) for i in range(nsets)]
    cols = [r(1620 + i) for i in range(ncols)]
    sig = [r(1640 + i) for i in range(ncols)]
    l0, ll, la, start = r(1660), r(1661), r(1662), r(1663)
    beta, gamma, y = O.random_fr(1671, 1), O.random_fr(1672, 1), O.random_fr(1673, 1)
    want = O.quotient_permutation(start, zs, cols, sig, chunk_len, l0, ll, la, beta, gamma, y, k, ext_k, 6)
    got = quotient_permutation(dev(start), [dev(z) for z in zs], [dev(c) for c in cols], [dev(s) for s in sig],
                               chunk_len, dev(l0), dev(ll), dev(la), beta, gamma, y, k, ext_k, 6)
    assert (got.cpu().numpy() == want).all()


@pytest.mark.parametrize("k,ext_k", [(4, 6), (10, 13), (13, 15)])
def test_quotient_lookup_random_parity(gpu, O, k, ext_k):
    from circuits_halo2_amd.arithmetic import quotient_lookup
    ne = 1 << ext_k
    arrs = [O.random_fr(1700 + i, ne) for i in range(9)]
    beta, gamma, y = O.random_fr(1711, 1), O.random_fr(1712, 1), O.random_fr(1713, 1)
    want = O.quotient_lookup(arrs[0], *arrs[1:], beta, gamma, y, k, ext_k)
    got = quotient_lookup(*[dev(a) for a in arrs], beta, gamma, y, k, ext_k)
    assert (got.cpu().numpy() == want).all()


def test_quotient_shape_errors(gpu, O):
    from circuits_halo2_amd.arithmetic import quotient_permutation
    from circuits_halo2_amd.ffi import SummaGpuError
    k, ext_k = 4, 6
    t = lambda: dev(O.random_fr(1, 1 << ext_k))
    b = O.random_fr(2, 1)
    with pytest.raises(SummaGpuError):           # 3 columns in chunks of 2 need exactly 2 sets
        quotient_permutation(t(), [t()], [t(), t(), t()], [t(), t(), t()], 2, t(), t(), t(), b, b, b, k, ext_k, 6)
    with pytest.raises(ValueError):
        quotient_permutation(t(), [t()], [t()], [t(), t()], 2, t(), t(), t(), b, b, b, k, ext_k, 6)
    with pytest.raises(ValueError):
        quotient_permutation(t()[:64], [t()], [t()], [t()], 2, t(), t(), t(), b, b, b, k, ext_k, 6)


def _pipeline_inputs(O, k, ncols, chunk_len):
    """a witness that satisfies a permutation argument, a lookup argument and one custom gate, as Lagrange columns, their
    extended-coset evaluations on the device, and the gate's program"""
    import quotient_witness as W
    blinding, n = 5, 1 << k
    ext_k = k + 3 if chunk_len == 4 else k + 2                 # degree 6 (MstInclusion's) / degree 4
    ne = 1 << ext_k
    beta, gamma, y = O.random_fr(1801, 1), O.random_fr(1802, 1), O.random_fr(1803, 1)
    u, l0, l_last, l_active = W.selectors(k, blinding)
    cols, sigmas, zs = W.permutation_witness(k, ncols, chunk_len, blinding, 1810 + k, beta, gamma)
    a, s, ap, sp, z = W.lookup_witness(k, blinding, 1820 + k, beta, gamma)
    # a custom gate  q * (ga * gb(omega X) - gc) = 0  on the usable rows, degree 3
    ga, gb = O.random_fr(1830 + k, n), O.random_fr(1831 + k, n)
    gb_next = np.roll(gb.reshape(n, 32), -1, axis=0).reshape(-1)
    gc = O.fr_mul_n(ga, gb_next)
    gc[32 * u:] = O.random_fr(1832 + k, n - u)                       # blinding rows: anything
    gq = l_active.copy()
    lag = [l0, l_last, l_active, *zs, *cols, *sigmas, z, ap, sp, a, s, ga, gb, gc, gq]
    ext, dom = _to_extended_dev(lag, k, ext_k)
    e_l0, e_ll, e_la = ext[:3]
    e_zs = ext[3:3 + len(zs)]
    e_cols = ext[3 + len(zs):3 + len(zs) + ncols]
    e_sig = ext[3 + len(zs) + ncols:3 + len(zs) + 2 * ncols]
    e_z, e_ap, e_sp, e_a, e_s = ext[-9:-4]
    e_ga, e_gb, e_gc, e_gq = ext[-4:]
    from circuits_halo2_amd import arithmetic as A
    theta = O.random_fr(1804, 1)
    graph = A.GraphEvaluator()
    prod = graph.add_calculation(A.MUL, graph.query(A.ADVICE, 0, 0), graph.query(A.ADVICE, 1, 1))
    gate = graph.add_calculation(A.MUL, graph.query(A.FIXED, 0, 0),
                                 graph.add_calculation(A.SUB, prod, graph.query(A.ADVICE, 2, 0)))
    graph.add_calculation(A.HORNER, (A.PREVIOUS_VALUE, 0, 0), (A.Y, 0, 0), [gate])
    return locals()


@pytest.mark.parametrize("k,ncols,chunk_len", [(8, 6, 2), (12, 6, 4)])
def test_quotient_pipeline_satisfying_witness(gpu, O, k, ncols, chunk_len):
    """whole h(X) pipeline on the device with a witness that satisfies both arguments: grand products
    -> iNTT -> coset NTT -> numerator -> / (X^n - 1) -> coset iNTT; the quotient is a polynomial of the
    expected degree (top coefficients vanish), identical to the oracle's, and a tampered witness is caught"""
    import torch
    import quotient_witness as W
    from circuits_halo2_amd import arithmetic as A
    from circuits_halo2_amd.arithmetic import quotient_lookup, quotient_permutation
    ctx = _pipeline_inputs(O, k, ncols, chunk_len)
    (blinding, n, ext_k, ne, beta, gamma, y, theta, graph, dom, l0, l_last, l_active, zs, cols, sigmas, z, ap, sp, a, s, ga, gb, gc, gq,
     e_l0, e_ll, e_la, e_zs, e_cols, e_sig, e_z, e_ap, e_sp, e_a, e_s, e_ga, e_gb, e_gc, e_gq) = (ctx[v] for v in (
        "blinding", "n", "ext_k", "ne", "beta", "gamma", "y", "theta", "graph", "dom", "l0", "l_last", "l_active", "zs", "cols", "sigmas",
        "z", "ap", "sp", "a", "s", "ga", "gb", "gc", "gq", "e_l0", "e_ll", "e_la", "e_zs", "e_cols", "e_sig", "e_z", "e_ap", "e_sp",
        "e_a", "e_s", "e_ga", "e_gb", "e_gc", "e_gq"))
    values = torch.zeros(32 * ne, dtype=torch.uint8, device="cuda")
    gate_args = ([e_gq], [e_ga, e_gb, e_gc], [], np.zeros(0, dtype=np.uint8), beta, gamma, theta, y, k, ext_k)
    A.quotient_gates(values, graph, *gate_args)                       # halo2's order: gates, permutation, lookups
    assert values.any()
    quotient_permutation(values, e_zs, e_cols, e_sig, chunk_len, e_l0, e_ll, e_la, beta, gamma, y, k, ext_k, blinding + 1)
    quotient_lookup(values, e_z, e_ap, e_sp, e_a, e_s, e_l0, e_ll, e_la, beta, gamma, y, k, ext_k)
    # same folds by the oracle on the same extended columns
    h = lambda t: t.cpu().numpy()
    want = O.quotient_gates(np.zeros(32 * ne, dtype=np.uint8), graph.as_dict(), [h(e_gq)], [h(e_ga), h(e_gb), h(e_gc)], [],
                            *gate_args[3:])
    want = O.quotient_permutation(want, [h(t) for t in e_zs], [h(t) for t in e_cols],
                                  [h(t) for t in e_sig], chunk_len, h(e_l0), h(e_ll), h(e_la), beta, gamma, y, k, ext_k,
                                  blinding + 1)
    want = O.quotient_lookup(want, h(e_z), h(e_ap), h(e_sp), h(e_a), h(e_s), h(e_l0), h(e_ll), h(e_la), beta, gamma, y,
                             k, ext_k)
    assert (h(values) == want).all()
    full = values.clone()
    dom.divide_by_vanishing_poly(full)
    ffi_check_full = dom.extended_to_coeff(full)                # truncated view; `full` now holds all coefficients
    deg = max(chunk_len + 2, 4)
    first_zero = (deg - 1) * n - deg + 1
    assert W.top_coefficients_zero(h(full), first_zero)
    assert h(ffi_check_full).any()
    # the verifier's equation at a random point: h(x) (x^n - 1) == numerator(x), the right-hand side from the
    # single-point formulas of the reference's generated verifier (W.verifier_numerator, Python integers) on
    # evaluations of the committed polynomials -- ties the extended-coset pipeline to the verifier's view
    from oracle import pyref as PR
    toi = lambda b: PR.fr_from_bytes(bytes(b))
    xi = PR.random_fr(4242 + k, 1)[0]
    w = PR.omega_for(k)
    pts = {"x": xi, "next": xi * w % PR.R, "prev": xi * pow(w, -1, PR.R) % PR.R,
           "last": xi * pow(w, -(blinding + 1), PR.R) % PR.R}
    co = lambda col: O.lagrange_to_coeff(col, k)
    ev_at = lambda coeffs, pt: toi(O.fr_eval_poly(coeffs, fr_np([pts[pt]])))
    c_z = [co(t) for t in zs]
    c_lz, c_ap = co(z), co(ap)
    ev = {"l0": ev_at(co(l0), "x"), "l_last": ev_at(co(l_last), "x"), "l_active": ev_at(co(l_active), "x"),
          "z": [ev_at(c, "x") for c in c_z], "z_next": [ev_at(c, "next") for c in c_z],
          "z_last": [ev_at(c, "last") for c in c_z],
          "cols": [ev_at(co(c), "x") for c in cols], "sigma": [ev_at(co(c), "x") for c in sigmas],
          "lz": ev_at(c_lz, "x"), "lz_next": ev_at(c_lz, "next"), "ap": ev_at(c_ap, "x"), "ap_prev": ev_at(c_ap, "prev"),
          "sp": ev_at(co(sp), "x"), "a": ev_at(co(a), "x"), "s": ev_at(co(s), "x"),
          "gq": ev_at(co(gq), "x"), "ga": ev_at(co(ga), "x"), "gb_next": ev_at(co(gb), "next"), "gc": ev_at(co(gc), "x")}
    num_x = W.verifier_numerator(ev, toi(beta), toi(gamma), toi(y), xi, chunk_len)
    h_x = toi(O.fr_eval_poly(h(full), fr_np([xi])))
    assert h_x * (pow(xi, n, PR.R) - 1) % PR.R == num_x
    # tamper: one cell of one permutation column
    bad = cols[1].copy()
    bad[32 * 3:32 * 4] = O.fr_add(bad[32 * 3:32 * 4].copy(), W.fr_np([1]))
    e_bad = dom.coeff_to_extended(dom.lagrange_to_coeff(dev(bad)))
    values = torch.zeros(32 * ne, dtype=torch.uint8, device="cuda")
    quotient_permutation(values, e_zs, [e_cols[0], e_bad, *e_cols[2:]], e_sig, chunk_len, e_l0, e_ll, e_la, beta, gamma,
                         y, k, ext_k, blinding + 1)
    dom.divide_by_vanishing_poly(values)
    dom.extended_to_coeff(values)
    assert not W.top_coefficients_zero(h(values), first_zero)
    # tamper: one cell of the gate's output column
    bad = gc.copy()
    bad[32 * 5:32 * 6] = O.fr_add(bad[32 * 5:32 * 6].copy(), W.fr_np([1]))
    e_bad = dom.coeff_to_extended(dom.lagrange_to_coeff(dev(bad)))
    values = torch.zeros(32 * ne, dtype=torch.uint8, device="cuda")
    A.quotient_gates(values, graph, [e_gq], [e_ga, e_gb, e_bad], [], *gate_args[3:])
    dom.divide_by_vanishing_poly(values)
    dom.extended_to_coeff(values)
    assert not W.top_coefficients_zero(h(values), first_zero)


@pytest.mark.parametrize("k", [10, 19])
def test_cosets_round_trip_from_known_pieces(gpu, O, k):
    """pieces h_0..h_4 (random) -> the numerator h (X^n - 1) on the five cosets, built block by block from the coset transforms of
    the pieces (on c_b H: X^n = g_b, so numerator_b = (g_b - 1) sum_t g_b^t h_t) -> sg_cosets_to_pieces gives the pieces back.
    k = 10 runs the batched transforms that the oracle-based tests cover; k = 19 the one-by-one path of long transforms, and
    the de-interleaving identity against coeff_to_extended there too."""
    import torch
    from circuits_halo2_amd import arithmetic as A
    from circuits_halo2_amd.domain import EvaluationDomain
    from oracle import pyref as PR
    dom = EvaluationDomain(6, k)
    n, d, ext_k = 1 << k, dom.quotient_poly_degree, dom.extended_k
    pieces = [dev(O.random_fr(3100 + t, n)) for t in range(d)]
    cos = dom.coeff_to_cosets_batch(pieces)
    w_ext = PR.omega_for(ext_k)
    gam = [pow(PR.ZETA * pow(w_ext, b, PR.R) % PR.R, n, PR.R) for b in range(d)]
    assert len(set(gam)) == d and 1 not in gam
    blocks = []
    for b in range(d):
        coeffs = fr_np([(gam[b] - 1) * pow(gam[b], t, PR.R) % PR.R for t in range(d)])
        blocks.append(A.lincomb([c[32 * n * b:32 * n * (b + 1)] for c in cos], coeffs))
    back = dom.cosets_to_pieces(torch.cat(blocks))
    for t in range(d):
        assert (back[t] == pieces[t]).all(), t
    ext = dom.coeff_to_extended(pieces[0].clone())
    stride = 1 << (ext_k - k)
    assert (cos[0] == ext.view(n, stride, 32)[:, :d, :].permute(1, 0, 2).reshape(-1)).all()


@pytest.mark.parametrize("k,ncols,chunk_len", [(6, 6, 4), (8, 6, 2), (12, 6, 4)])
def test_quotient_on_cosets_equals_the_extended_pipeline(gpu, O, k, ncols, chunk_len):
    """the prover's short cut (sg_coeff_to_cosets / sg_cosets_to_pieces): deg h < d n, so the quotient is computed on the first d =
    degree - 1 cosets of the extended domain only.  (1) the coset-major transform is the de-interleaved coeff_to_extended
    (oracle-checked elsewhere), column by column; (2) gates / permutation / lookup folded coset by coset give the extended
    numerator's rows; (3) the pieces equal extended_to_coeff(divide_by_vanishing_poly(numerator)) coefficient for
    coefficient."""
    import torch
    from circuits_halo2_amd import arithmetic as A
    from circuits_halo2_amd.domain import EvaluationDomain
    ctx = _pipeline_inputs(O, k, ncols, chunk_len)
    n, ext_k, ne, blinding = ctx["n"], ctx["ext_k"], ctx["ne"], ctx["blinding"]
    beta, gamma, y, theta, graph = ctx["beta"], ctx["gamma"], ctx["y"], ctx["theta"], ctx["graph"]
    deg = max(chunk_len + 2, 4)
    dom = EvaluationDomain(deg, k)
    d, stride = dom.quotient_poly_degree, 1 << (ext_k - k)
    assert dom.extended_k == ext_k and d == deg - 1
    names = ["l0", "l_last", "l_active", "z", "ap", "sp", "a", "s", "ga", "gb", "gc", "gq"]
    lag = [ctx[v] for v in names] + list(ctx["zs"]) + list(ctx["cols"]) + list(ctx["sigmas"])
    ext = [ctx[v] for v in ("e_l0", "e_ll", "e_la", "e_z", "e_ap", "e_sp", "e_a", "e_s", "e_ga", "e_gb", "e_gc", "e_gq")] + \
        list(ctx["e_zs"]) + list(ctx["e_cols"]) + list(ctx["e_sig"])
    coeffs = [dom.lagrange_to_coeff(dev(c)) for c in lag]
    cos = dom.coeff_to_cosets_batch(coeffs)
    for cm, e in zip(cos, ext):                                   # (1)
        want = e.view(n, stride, 32)[:, :d, :].permute(1, 0, 2).reshape(-1)
        assert (cm == want).all()
    assert dom.coeff_to_cosets_batch([]) == []
    c = dict(zip(names, cos[:len(names)]))
    nz = len(ctx["zs"])
    c_zs, c_cols, c_sig = cos[len(names):len(names) + nz], cos[len(names) + nz:len(names) + nz + ncols], cos[len(names) + nz + ncols:]
    values = torch.zeros(32 * d * n, dtype=torch.uint8, device="cuda")
    none = np.zeros(0, dtype=np.uint8)
    A.quotient_gates_cosets(values, graph, [c["gq"]], [c[t] for t in ("ga", "gb", "gc")], [], none, beta, gamma, theta, y, k, d)   # (2)
    A.quotient_permutation_cosets(values, c_zs, c_cols, c_sig, chunk_len, c["l0"], c["l_last"], c["l_active"], beta, gamma, y, k, ext_k, d,
                                  blinding + 1)
    A.quotient_lookup_cosets(values, *[c[t] for t in ("z", "ap", "sp", "a", "s", "l0", "l_last", "l_active")], beta, gamma, y, k, d)
    # a single block through the per-domain entry points (2^k rows, ext_k = k) gives the same rows: gates and lookup do not see the coset
    blk = lambda t, b: t[32 * n * b:32 * n * (b + 1)].clone()
    one = torch.zeros(32 * n, dtype=torch.uint8, device="cuda")
    A.quotient_gates(one, graph, [blk(c["gq"], 1)], [blk(c[t], 1) for t in ("ga", "gb", "gc")], [], none, beta, gamma, theta, y, k, k)
    chk = torch.zeros(32 * d * n, dtype=torch.uint8, device="cuda")
    A.quotient_gates_cosets(chk, graph, [c["gq"]], [c[t] for t in ("ga", "gb", "gc")], [], none, beta, gamma, theta, y, k, d)
    assert (one == blk(chk, 1)).all()
    with pytest.raises(ValueError):
        A.quotient_gates_cosets(values[:64], graph, [c["gq"]], [c[t] for t in ("ga", "gb", "gc")], [], none, beta, gamma, theta, y, k, d)
    full = torch.zeros(32 * ne, dtype=torch.uint8, device="cuda")
    A.quotient_gates(full, graph, [ctx["e_gq"]], [ctx["e_ga"], ctx["e_gb"], ctx["e_gc"]], [], np.zeros(0, dtype=np.uint8), beta, gamma,
                     theta, y, k, ext_k)
    A.quotient_permutation(full, ctx["e_zs"], ctx["e_cols"], ctx["e_sig"], chunk_len, ctx["e_l0"], ctx["e_ll"], ctx["e_la"], beta, gamma,
                           y, k, ext_k, blinding + 1)
    A.quotient_lookup(full, ctx["e_z"], ctx["e_ap"], ctx["e_sp"], ctx["e_a"], ctx["e_s"], ctx["e_l0"], ctx["e_ll"], ctx["e_la"], beta,
                      gamma, y, k, ext_k)
    assert (values == full.view(n, stride, 32)[:, :d, :].permute(1, 0, 2).reshape(-1)).all()
    ctx["dom"].divide_by_vanishing_poly(full)                     # (3)
    want = ctx["dom"].extended_to_coeff(full)[:32 * d * n].clone()
    pieces = dom.cosets_to_pieces(values)
    assert len(pieces) == d and want.any()
    assert (torch.cat(pieces) == want).all()
    with pytest.raises(ValueError):
        dom.cosets_to_pieces(values[:64])


# ----------------------------------------------------------------------------- witness side (row W)
def test_k5_merkle_sum_tree_on_gpu(gpu, kat, P):
    """the reference's own constants: entry_16.csv -> leaf 0, leaf 1, root hash and root balances
    (zk_prover/src/circuits/tests.rs:341,346; backend/src/tests.rs:265,268)"""
    import os
    from conftest import GOLDEN
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    tree = MerkleSumTree.from_csv(os.path.join(GOLDEN, "entry_16.csv"), 2)
    assert tree.depth == 4
    assert P.fr_from_bytes(tree.node(0, 0)[0].tobytes()) == int(kat["k5"]["leaf0"], 16)
    assert P.fr_from_bytes(tree.node(0, 1)[0].tobytes()) == int(kat["k5"]["leaf1"], 16)
    rh, rb = tree.root()
    assert P.fr_from_bytes(rh.tobytes()) == int(kat["k5"]["root"], 16)
    assert P.frs_from_bytes(rb.tobytes()) == kat["k5"]["root_balances"]
    # Merkle proof of user 0 recomputes the root with the oracle's middle-node hash
    proof = tree.generate_proof(0)
    node = (P.fr_from_bytes(proof["leaf"][0].tobytes()), P.frs_from_bytes(proof["leaf"][1].tobytes()))
    for (sh, sb), bit in zip(proof["siblings"], proof["path_indices"]):
        sib = (P.fr_from_bytes(sh.tobytes()), P.frs_from_bytes(sb.tobytes()))
        node = P.mst_middle(sib, node) if bit else P.mst_middle(node, sib)
    assert node[0] == int(kat["k5"]["root"], 16)


def test_merkle_sum_tree_reference_api(gpu, P):
    """the rest of the reference's MerkleSumTree / Tree API on the device-hashed tree: proofs with sibling
    preimages verify (tree.rs:85-190), a tampered proof does not, update_leaf (mst.rs:169-204) equals a rebuild,
    from_csv_sorted + index_of_username (mst.rs:89-100, 207-223)"""
    import os
    from conftest import GOLDEN
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree, parse_csv_to_entries
    path = os.path.join(GOLDEN, "entry_16.csv")
    tree = MerkleSumTree.from_csv(path, 2)
    entries, crypto = parse_csv_to_entries(path, 2)
    assert len(crypto) == 2
    for i in (0, 1, 7, 15):
        proof = tree.generate_proof(i)
        assert proof["entry"][0] == entries[i][0]
        assert len(proof["sibling_middle_node_hash_preimages"]) == tree.depth - 1
        assert tree.verify_proof(proof)
    bad = tree.generate_proof(3)
    bad["entry"] = (bad["entry"][0], [bad["entry"][1][0] + 1, bad["entry"][1][1]])
    assert not tree.verify_proof(bad)
    bad = tree.generate_proof(3)
    bad["path_indices"][1] ^= 1
    assert not tree.verify_proof(bad)
    with pytest.raises(IndexError):
        tree.generate_proof(16)
    # update_leaf == rebuild from the modified entries == the big-integer tree
    name = entries[5][0]
    assert tree.index_of_username(name) == 5
    new_root = tree.update_leaf(name, [123456, 7])
    entries[5] = (name, [123456, 7])
    rebuilt = MerkleSumTree.from_entries(entries, 2)
    assert (new_root[0] == rebuilt.root()[0]).all() and (new_root[1] == rebuilt.root()[1]).all()
    assert (tree._h == rebuilt._h).all() and (tree._b == rebuilt._b).all()
    ref = P.mst_build([P.mst_entry(n, b) for n, b in entries])
    assert P.fr_from_bytes(new_root[0].tobytes()) == ref[0][0]
    assert tree.verify_proof(tree.generate_proof(5))
    with pytest.raises(KeyError):
        tree.index_of_username("nobody")
    # sorted build: binary search finds every user at the position of the sorted order
    st = MerkleSumTree.from_csv_sorted(path, 2)
    order = sorted(e[0] for e in entries)
    for pos, nm in enumerate(order):
        assert st.index_of_username(nm) == pos
    assert st.verify_proof(st.generate_proof(st.index_of_username(order[3])))
    # a tree that needs padding: 13 users -> 16 leaves; the proof of the last real user has a zero-entry sibling
    t13 = MerkleSumTree.from_entries(entries[:13], 2)
    assert t13.depth == 4 and t13.verify_proof(t13.generate_proof(12))


@pytest.mark.parametrize("n,nc", [(1, 1), (13, 2), (17, 2), (1000, 3), (1 << 14, 2)])
def test_merkle_sum_tree_vs_oracle(gpu, O, n, nc):
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree, keccak256
    rng = np.random.default_rng(n)
    entries = [("user%d" % i, [int(v) for v in rng.integers(0, 1 << 60, nc)]) for i in range(n)]
    tree = MerkleSumTree.from_entries(entries, nc)
    size = 1 << tree.depth
    users = fr_np([int.from_bytes(keccak256(u.encode()), "big") for u, _ in entries] + [0] * (size - n))
    bals = fr_np([v for _, b in entries for v in b] + [0] * (nc * (size - n)))
    h, b = O.mst_leaves(users, bals, nc), bals
    assert (tree._h[:32 * size] == h).all()
    off = size
    m = size >> 1
    while m >= 1:
        h, b = O.mst_level(h, b, nc)
        assert (tree._h[32 * off:32 * (off + m)] == h).all()
        assert (tree._b[32 * off * nc:32 * (off + m) * nc] == b).all()
        off += m
        m >>= 1
    # the tree takes any BigUint balance (mst.rs:103-134; only the circuit's range check objects to one beyond N_BYTES:
    # circuits/tests.rs:268-299); a negative one cannot exist
    big = MerkleSumTree.from_entries([("x", [1 << 64] * nc)], nc, n_bytes=8)
    assert bytes(big.node(0, 0)[1]) == fr_np([1 << 64] * nc).tobytes()
    with pytest.raises(ValueError):
        MerkleSumTree.from_entries([("x", [-1] * nc)], nc, n_bytes=8)


# ---------------------------------------------------------------- the reference circuit's own constraint system
@pytest.mark.parametrize("k,ext_k,sample", [(5, 8, 0), (11, 14, 96)])
def test_mst_inclusion_constraint_system_on_device(gpu, O, k, ext_k, sample):
    """evaluate_h's custom-gate block with the constraint system of the reference's MstInclusionCircuit
    (circuits_halo2_amd.mst_inclusion; k = 11 / 2^14 extended rows is the reference's own configuration):
    bit-exact against (a) the C oracle's interpreter of the same program on every row and (b) the gate polynomials
    of the restated verifier -- the one that accepts the reference's shipped proof, tests/test_verifier_cpu.py --
    evaluated with Python integers row by row (every row at k = 5, a sample at k = 11).  Same for the lookup's
    compressed input expression."""
    from circuits_halo2_amd import arithmetic as A
    from circuits_halo2_amd import mst_inclusion as M
    from oracle import pyref as PR
    from oracle import summa_verifier as SV
    ne, step = 1 << ext_k, 1 << (ext_k - k)
    fixed = [O.random_fr(5100 + i, ne) for i in range(M.NUM_FIXED)]
    advice = [O.random_fr(5200 + i, ne) for i in range(M.NUM_ADVICE)]
    beta, gamma, theta, y = (O.random_fr(5300 + i, 1) for i in range(4))
    start = O.random_fr(5310, ne)
    none = np.zeros(0, dtype=np.uint8)
    g = M.gate_graph()
    chal = M.gate_challenges(PR.fr_from_bytes(bytes(y)))     # the powers of y the factored program reads
    want = O.quotient_gates(start, g.as_dict(), fixed, advice, [], chal, beta, gamma, theta, y, k, ext_k)
    got = A.quotient_gates(dev(start), g, [dev(c) for c in fixed], [dev(c) for c in advice], [], chal, beta, gamma, theta, y,
                           k, ext_k).cpu().numpy()
    assert (got == want).all()
    gi = M.expression_graph(M.lookup_expressions()[0])
    got_in = A.quotient_gates(dev(start), gi, [dev(c) for c in fixed], [dev(c) for c in advice], [], none, beta, gamma, theta,
                              y, k, ext_k).cpu().numpy()
    cell = lambda arr, row: PR.fr_from_bytes(bytes(arr[32 * row:32 * row + 32]))
    yi = PR.fr_from_bytes(bytes(y))
    rows = range(ne) if not sample else [int(r) for r in np.random.default_rng(7).integers(0, ne, sample)] + [0, ne - 1]
    for row in rows:
        q = lambda kind, c, rot: cell((fixed if kind == "f" else advice)[c], (row + rot * step) % ne)
        acc = cell(start, row)
        for t in SV.gate_values(q):
            acc = (acc * yi + t) % PR.R
        assert cell(got, row) == acc, row
        assert cell(got_in, row) == SV.lookup_input_table(q)[0], row
