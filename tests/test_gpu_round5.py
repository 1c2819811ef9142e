"""GPU tests of round 5's kernels, all through the C ABI: the one-pass quotient numerator against the separate kernels (which the
oracle pins, tests/test_gpu_parity.py), the coset transform with the shift folded into the first NTT pass against the pass of
its own, the rotation sets' linear combinations in one launch against one launch each, and the profiling / parameter entry
points (launch log, sg_get_param, ABI revision)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    import circuits_halo2_amd as sg
    from circuits_halo2_amd import ffi
    ffi.check(sg.lib().sg_init(0))
    return sg


def _rand_fr(seed, rows):
    """`rows` canonical Montgomery field elements on the device"""
    import torch
    from circuits_halo2_amd.arithmetic import fr_to_montgomery
    from circuits_halo2_amd.utils import random_fr_canonical
    return fr_to_montgomery(torch.from_numpy(random_fr_canonical(seed, rows)).cuda())


def _scalar(seed):
    return _rand_fr(seed, 1).cpu().numpy()


@pytest.mark.parametrize("nc", [1, 2, 3, 4])
@pytest.mark.parametrize("k", [5, 9])
def test_fused_numerator_matches_the_separate_kernels(gpu, nc, k):
    """sg_quotient_numerator_cosets_dev with the reference circuit's own programs (the fused kernel) writes the words the separate
    gate / permutation / lookup kernels write -- which the oracle pins -- over random columns; `values` arrives full of
    garbage (the entry point promises that it needs no clearing); k = 5: blocks shorter than a workgroup (every thread takes
    its own coset's shift), k = 9: the uniform path"""
    import torch
    from circuits_halo2_amd import arithmetic as A, ffi, mst_inclusion as M
    d, n = 5, 1 << k
    ext_k = k + 3
    rows = d * n
    seed = iter(range(1000 * nc + 100 * k, 10 ** 6))
    col = lambda: _rand_fr(next(seed), rows)
    fixed = [col() for _ in range(M.NUM_FIXED)]
    advice = [col() for _ in range(M.NUM_ADVICE)]
    inst = [col()]
    zs = [col(), col()]
    sigmas = [col() for _ in range(6)]
    perm_cols = [fixed[2], advice[0], advice[1], fixed[3], advice[2], inst[0]]
    l0, l_last, l_active, lz, pin, ptab = col(), col(), col(), col(), col(), col()
    beta, gamma, theta, y = (_scalar(next(seed)) for _ in range(4))
    n_chal = len(M.gate_challenge_exponents(nc))
    challenges = _rand_fr(next(seed), n_chal).cpu().numpy()
    gates, look = M.gate_graph(nc), M.lookup_input_graph()
    none = np.zeros(0, dtype=np.uint8)

    def run(fused):
        ffi.set_param("quotient.fused_numerator", 1 if fused else 0)
        try:
            values = _rand_fr(4242, rows)         # garbage in
            A.quotient_numerator_cosets(values, gates, look, fixed, advice, inst, challenges, zs, perm_cols, sigmas, 4, l0, l_last, l_active,
                                        lz, pin, ptab, fixed[4], beta, gamma, theta, y, k, ext_k, d, 6)
            torch.cuda.synchronize()
            return values
        finally:
            ffi.set_param("quotient.fused_numerator", 1)

    got, want = run(True), run(False)
    assert (got == want).all()
    # ... and the separate kernels driven by hand (what the Python driver still does)
    values = torch.zeros(32 * rows, dtype=torch.uint8, device="cuda")
    A.quotient_gates_cosets(values, gates, fixed, advice, inst, challenges, beta, gamma, theta, y, k, d)
    A.quotient_permutation_cosets(values, zs, perm_cols, sigmas, 4, l0, l_last, l_active, beta, gamma, y, k, ext_k, d, 6)
    input_c = torch.empty(32 * rows, dtype=torch.uint8, device="cuda")
    A.quotient_gates_cosets(input_c, look, fixed, advice, inst, none, beta, gamma, theta, y, k, d)
    A.quotient_lookup_cosets(values, lz, pin, ptab, input_c, fixed[4], l0, l_last, l_active, beta, gamma, y, k, d)
    assert (got == values).all() and got.any()


def test_numerator_of_an_unknown_program_takes_the_separate_kernels(gpu):
    """a gate program the library has no straight-line code for (here: the lookup's input expression in the gates' place) runs the
    separate kernels behind the same entry point, fused or not"""
    import torch
    from circuits_halo2_amd import arithmetic as A, ffi, mst_inclusion as M
    k, d = 6, 5
    rows = d << k
    seed = iter(range(777, 10 ** 6))
    col = lambda: _rand_fr(next(seed), rows)
    fixed, advice, inst = [col() for _ in range(M.NUM_FIXED)], [col() for _ in range(M.NUM_ADVICE)], [col()]
    zs, sigmas = [col(), col()], [col() for _ in range(6)]
    perm_cols = [fixed[2], advice[0], advice[1], fixed[3], advice[2], inst[0]]
    l0, l_last, l_active, lz, pin, ptab = col(), col(), col(), col(), col(), col()
    beta, gamma, theta, y = (_scalar(next(seed)) for _ in range(4))
    look = M.lookup_input_graph()
    none = np.zeros(0, dtype=np.uint8)
    got = _rand_fr(1, rows)
    A.quotient_numerator_cosets(got, look, look, fixed, advice, inst, none, zs, perm_cols, sigmas, 4, l0, l_last, l_active, lz, pin, ptab,
                                fixed[4], beta, gamma, theta, y, k, k + 3, d, 6)
    values = torch.zeros(32 * rows, dtype=torch.uint8, device="cuda")
    A.quotient_gates_cosets(values, look, fixed, advice, inst, none, beta, gamma, theta, y, k, d)
    A.quotient_permutation_cosets(values, zs, perm_cols, sigmas, 4, l0, l_last, l_active, beta, gamma, y, k, k + 3, d, 6)
    input_c = torch.empty(32 * rows, dtype=torch.uint8, device="cuda")
    A.quotient_gates_cosets(input_c, look, fixed, advice, inst, none, beta, gamma, theta, y, k, d)
    A.quotient_lookup_cosets(values, lz, pin, ptab, input_c, fixed[4], l0, l_last, l_active, beta, gamma, y, k, d)
    assert (got == values).all()


@pytest.mark.parametrize("k,count", [(4, 1), (11, 3), (12, 2), (13, 7), (17, 5)])
def test_coset_shift_inside_the_first_ntt_pass(gpu, k, count):
    """sg_coeff_to_cosets_batch_dev: the shift c_b^i folded into the load of the first NTT pass (default) gives the words of the
    pass of its own (rounds 3-4, oracle-pinned through test_quotient_on_cosets_equals_the_extended_pipeline); 7 columns x 5
    cosets = 35 blocks: more than one batched launch holds"""
    import torch
    from circuits_halo2_amd import ffi
    from circuits_halo2_amd.domain import EvaluationDomain
    dom = EvaluationDomain(6, k)
    coeffs = [_rand_fr(31 * k + j, 1 << k) for j in range(count)]
    keep = [c.clone() for c in coeffs]

    def run(own_pass):
        ffi.set_param("ntt.coset_scale_pass", own_pass)
        try:
            out = dom.coeff_to_cosets_batch(coeffs)
            torch.cuda.synchronize()
            return out
        finally:
            ffi.set_param("ntt.coset_scale_pass", 0)

    folded, separate = run(0), run(1)
    assert len(folded) == count
    for a, b in zip(folded, separate):
        assert (a == b).all() and a.any()
    for c, c0 in zip(coeffs, keep):
        assert (c == c0).all()          # the inputs are read only


def test_lincomb_sets_is_the_single_combinations(gpu):
    """sg_fr_lincomb_sets_dev: five combinations of different sizes (one of them of 25 terms, one with no low polynomial) in one
    launch = sg_fr_lincomb_low_dev one at a time"""
    import ctypes as C
    import torch
    from circuits_halo2_amd import arithmetic as A, ffi
    n = 1 << 10
    sizes, lows = [2, 25, 1, 2, 1], [3, 1, 3, 0, 2]
    seed = iter(range(9000, 10 ** 6))
    sets = []
    for m, nl in zip(sizes, lows):
        polys = [_rand_fr(next(seed), n) for _ in range(m)]
        coeffs = _rand_fr(next(seed), m).cpu().numpy()
        low = _rand_fr(next(seed), nl).cpu().numpy() if nl else None
        sets.append((polys, coeffs, low))
    outs = A.fr_lincomb_sets(sets, n)
    torch.cuda.synchronize()
    L = ffi.lib()
    for (polys, coeffs, low), got in zip(sets, outs):
        want = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
        pp = (C.c_void_p * len(polys))(*[p.data_ptr() for p in polys])
        ffi.check(L.sg_fr_lincomb_low_dev(pp, ffi.ptr(ffi.u8(coeffs)), C.c_uint32(len(polys)), C.c_size_t(n),
                                          ffi.ptr(ffi.u8(low)) if low is not None else None, C.c_uint32(len(low) // 32 if low is not None else 0),
                                          ffi.dev_ptr(want), ffi.current_stream_ptr()))
        torch.cuda.synchronize()
        assert (got == want).all() and got.any()
    # bad shapes are refused, not launched
    too_many = [([sets[1][0][0]] * 33, np.zeros(33 * 32, dtype=np.uint8), None)]
    with pytest.raises(ffi.SummaGpuError):
        A.fr_lincomb_sets(too_many, n)


def test_launch_log_get_param_and_abi_revision(gpu):
    """msm.acc_log: one record per msm_accumulate launch, in issue order, describing the job; sg_get_param returns what
    sg_set_param set; the library reports the ABI revision of the header it was built from"""
    import torch
    from circuits_halo2_amd import arithmetic as A, ffi
    L = ffi.lib()
    assert L.sg_abi_version() == 3
    n = 1 << 12
    scal = _rand_fr(5, n)
    bases = A.g1_fixed_base_mul(_rand_fr(6, n))
    ffi.set_param("msm.acc_log", 1)
    try:
        pts = [A.best_multiexp(scal, bases) for _ in range(3)]
        A.best_multiexp(scal[:32 * 100], bases[:64 * 100])
        log = ffi.msm_launch_log()
    finally:
        ffi.set_param("msm.acc_log", 0)
    assert all((p == pts[0]).all() for p in pts)
    assert [r["n"] for r in log][:3] == [n, n, n] and [r["n"] for r in log][3:] in ([], [100])
    assert all(r["M"] == 1 and r["fixed"] == 0 and r["threads"] > 0 for r in log)
    assert log[0]["entries"] % n == 0 and log[0]["task_len"] >= 4
    ffi.set_param("msm.acc_log", 1)          # setting it again clears the log
    assert ffi.msm_launch_log() == []
    ffi.set_param("msm.acc_log", 0)
    before = ffi.get_param("host.wait_sleep_us")
    ffi.set_param("host.wait_sleep_us", 37)
    assert ffi.get_param("host.wait_sleep_us") == 37
    ffi.set_param("host.wait_sleep_us", before)
    assert ffi.get_param("commit.combine_target") >= 1 and ffi.get_param("lanes") in range(1, 9)
    ffi.set_param("msm.log_seg", 5)
    assert ffi.get_param("msm.log_seg") == 5
    ffi.set_param("msm.log_seg", 0)
    with pytest.raises(ffi.SummaGpuError):
        ffi.get_param("no.such.parameter")


def test_batch_restores_the_parameters_it_found(gpu):
    """prove_batch's process-wide settings (combiner target / wait, sleeping waits) are put back to what the caller had set, not to
    built-in defaults (advisor, round 4)"""
    from circuits_halo2_amd import batch as B, ffi
    ffi.set_param("host.wait_sleep_us", 25)
    ffi.set_param("commit.combine_wait_us", 777)
    try:
        scope = B._ParamScope({"host.wait_sleep_us": 50, "commit.combine_wait_us": 5000})
        scope.enter()
        inner = B._ParamScope({"commit.combine_wait_us": 2000})
        inner.enter()
        assert ffi.get_param("commit.combine_wait_us") == 2000 and ffi.get_param("host.wait_sleep_us") == 50
        scope.leave()                       # the first batch ends while the second still runs: nothing is restored yet
        assert ffi.get_param("host.wait_sleep_us") == 50
        inner.leave()
        assert ffi.get_param("host.wait_sleep_us") == 25 and ffi.get_param("commit.combine_wait_us") == 777
    finally:
        ffi.set_param("host.wait_sleep_us", 0)
        ffi.set_param("commit.combine_wait_us", 300)


def _canon_small(values):
    """Montgomery device column of small integers"""
    import torch
    from circuits_halo2_amd.arithmetic import fr_to_montgomery
    a = np.zeros((len(values), 32), dtype=np.uint8)
    v = np.asarray(values, dtype=np.uint64)
    for b in range(4):
        a[:, b] = (v >> (8 * b)) & 0xff
    return fr_to_montgomery(torch.from_numpy(a.reshape(-1)).cuda())


def test_lookup_permutation_without_waits_matches_the_waiting_one(gpu):
    """sg_lookup_permute_small_async_dev: same A', S' (Montgomery words) as sg_lookup_permute_small_dev, call after call on one stream
    (every call cleans the work space of the next), a rejected input in between included; the verdict arrives in the status word"""
    import ctypes as C
    import torch
    from circuits_halo2_amd import ffi
    L = ffi.lib()
    rows = 3000
    rng = np.random.default_rng(5)
    table = _canon_small(np.arange(rows) % 256)
    status = torch.zeros(4, dtype=torch.int32, device="cuda")

    def both(values):
        inp = _canon_small(values)
        outs = [torch.zeros(32 * rows, dtype=torch.uint8, device="cuda") for _ in range(4)]
        status.zero_()
        ffi.check(L.sg_lookup_permute_small_async_dev(ffi.dev_ptr(inp), ffi.dev_ptr(table), C.c_size_t(rows), ffi.dev_ptr(outs[0]), ffi.dev_ptr(outs[1]),
                                                      C.c_void_p(status.data_ptr()), ffi.current_stream_ptr()))
        torch.cuda.synchronize()
        verdict = int(status[0].item())
        rc = L.sg_lookup_permute_small_dev(ffi.dev_ptr(inp), ffi.dev_ptr(table), C.c_size_t(rows), ffi.dev_ptr(outs[2]), ffi.dev_ptr(outs[3]),
                                           ffi.current_stream_ptr())
        return verdict, rc, outs

    for trial in range(3):
        verdict, rc, outs = both(rng.integers(0, 256, rows))
        assert verdict == 0 and rc == 0
        assert (outs[0] == outs[2]).all() and (outs[1] == outs[3]).all() and outs[0].any()
    bad = rng.integers(0, 256, rows)
    bad[17] = 300                                   # not in the table
    verdict, rc, _ = both(bad)
    assert verdict == 1 and rc != 0
    verdict, rc, outs = both(rng.integers(0, 256, rows))     # the call after a rejected one starts clean
    assert verdict == 0 and rc == 0 and (outs[0] == outs[2]).all() and (outs[1] == outs[3]).all()


def test_flag_noncanonical_closing_values_and_asynchronous_remainder(gpu):
    """the three small entry points that spare a proof its copy launches: a range check that only raises a flag, grand products that
    also leave z_p[usable] where the caller says, a Kate division whose remainder stays on the device"""
    import ctypes as C
    import torch
    from circuits_halo2_amd import ffi
    L = ffi.lib()
    k, n = 10, 1 << 10
    u = n - 6
    cols = [_rand_fr(40 + j, n) for j in range(3)]
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    pc = (C.c_void_p * 3)(*[c.data_ptr() for c in cols])
    ffi.check(L.sg_fr_flag_noncanonical_dev(pc, C.c_uint32(3), C.c_size_t(n), C.c_void_p(flag.data_ptr()), ffi.current_stream_ptr()))
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    cols[1][32 * 77:32 * 78] = 0xff                 # 2^256 - 1 >= r
    ffi.check(L.sg_fr_flag_noncanonical_dev(pc, C.c_uint32(3), C.c_size_t(n), C.c_void_p(flag.data_ptr()), ffi.current_stream_ptr()))
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    # grand products: two permutation chunks (4 + 2 columns) and one lookup, closing = z_p[u]
    vals = [_rand_fr(60 + j, n) for j in range(6)]
    sig = [_rand_fr(70 + j, n) for j in range(6)]
    look = [_rand_fr(80 + j, n) for j in range(4)]
    beta, gamma = _scalar(90), _scalar(91)
    pv, ps = (C.c_void_p * 6)(*[v.data_ptr() for v in vals]), (C.c_void_p * 6)(*[v.data_ptr() for v in sig])
    pl = (C.c_void_p * 4)(*[v.data_ptr() for v in look])
    chunk = (C.c_uint32 * 2)(4, 2)
    zs = [torch.zeros(32 * n, dtype=torch.uint8, device="cuda") for _ in range(3)]
    zs2 = [torch.zeros(32 * n, dtype=torch.uint8, device="cuda") for _ in range(3)]
    closing = torch.zeros(3 * 32, dtype=torch.uint8, device="cuda")
    pz, pz2 = (C.c_void_p * 3)(*[z.data_ptr() for z in zs]), (C.c_void_p * 3)(*[z.data_ptr() for z in zs2])
    ffi.check(L.sg_grand_products_closing_dev(pv, ps, chunk, C.c_uint32(2), pl, C.c_uint32(1), ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)),
                                              C.c_uint32(k), C.c_size_t(u), pz, C.c_void_p(closing.data_ptr()), ffi.current_stream_ptr()))
    ffi.check(L.sg_grand_products_dev(pv, ps, chunk, C.c_uint32(2), pl, C.c_uint32(1), ffi.ptr(ffi.u8(beta)), ffi.ptr(ffi.u8(gamma)),
                                      C.c_uint32(k), C.c_size_t(u), pz2, ffi.current_stream_ptr()))
    torch.cuda.synchronize()
    for p in range(3):
        assert (zs[p] == zs2[p]).all()
        assert (closing[32 * p:32 * p + 32] == zs[p][32 * u:32 * u + 32]).all() and closing[32 * p:32 * p + 32].any()
    # Kate division: the remainder on the device = the remainder handed back
    a, b = _rand_fr(95, n), _scalar(96)
    q1, q2 = torch.zeros(32 * n, dtype=torch.uint8, device="cuda"), torch.zeros(32 * n, dtype=torch.uint8, device="cuda")
    rem_dev = torch.zeros(32, dtype=torch.uint8, device="cuda")
    rem = np.zeros(32, dtype=np.uint8)
    ffi.check(L.sg_fr_kate_division_rem_dev(ffi.dev_ptr(a), C.c_size_t(n), ffi.ptr(ffi.u8(b)), ffi.dev_ptr(q1), ffi.dev_ptr(rem_dev), ffi.current_stream_ptr()))
    ffi.check(L.sg_fr_kate_division_dev(ffi.dev_ptr(a), C.c_size_t(n), ffi.ptr(ffi.u8(b)), ffi.dev_ptr(q2), ffi.ptr(rem), ffi.current_stream_ptr()))
    torch.cuda.synchronize()
    assert (q1 == q2).all() and (rem_dev.cpu().numpy() == rem).all() and rem.any()


def test_one_launch_msm_of_a_handful_of_points(gpu):
    """sg_msm_g1 with at most 64 points (the verifier's 37) is ONE launch that reads its inputs from mapped host memory
    (MsmEngine::run_tiny): against the oracle, against the engine's pipeline over the same inputs (device pointers, and the host
    entry with "msm.tiny_max" = 0), on every special case of the group law a bucket or the suffix scan can meet -- one point many
    times (P + P in a bucket, equal partial sums in the scan), P and -P cancelling, identity points, zero scalars, scalars at the
    top of the range and straddling window boundaries, words >= r"""
    import torch
    from conftest import fr_np, point_np
    from oracle import oracle as O, pyref as P
    from circuits_halo2_amd import ffi
    bases_all = O.fixed_base_mul(O.random_fr(77, 80), 4)          # 80 points x_i G
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def pipeline(sc, bs):
        assert ffi.get_param("msm.tiny_max") == 64
        ffi.set_param("msm.tiny_max", 0)
        try:
            return gpu.best_multiexp(sc, bs)
        finally:
            ffi.set_param("msm.tiny_max", 64)

    for n in (1, 2, 3, 7, 8, 9, 31, 37, 63, 64, 65):
        sc, bs = O.random_fr(500 + n, n), bases_all[:64 * n]
        want = O.best_multiexp(sc, bs, 1)
        assert (gpu.best_multiexp(sc, bs) == want).all(), n
        assert (pipeline(sc, bs) == want).all(), n
        assert (gpu.best_multiexp(dev(sc), dev(bs)) == want).all(), n
    ident = np.zeros(64, dtype=np.uint8)
    g = bases_all[:64]
    gi = P.g1_from_bytes(g.tobytes())
    neg = point_np(P.g1_neg(gi))
    rng = np.random.default_rng(5)
    for n in (16, 37, 64):
        same = np.tile(g, n)
        small = fr_np([int(v) for v in rng.integers(0, 4, size=n)])           # few distinct digits: deep equal buckets
        x = O.random_fr(77, 80)[:32]                                           # the base is x G: sum_i s_i (x G) = ((sum s_i) x) G
        scaled = lambda sc: O.fixed_base_mul(O.fr_dot(O.fr_dot(sc, fr_np([1] * n)), x), 1)
        assert (gpu.best_multiexp(small, same) == scaled(small)).all()
        rand = O.random_fr(900 + n, n)
        assert (gpu.best_multiexp(rand, same) == scaled(rand)).all()
        assert (gpu.best_multiexp(np.tile(rand[:32], n), same) == scaled(np.tile(rand[:32], n))).all()
        alt = np.concatenate([g, neg] * (n // 2))
        pairs = np.repeat(O.random_fr(950 + n, n // 2).reshape(-1, 32), 2, axis=0).reshape(-1)
        assert not gpu.best_multiexp(pairs, alt).any()
        assert not gpu.best_multiexp(fr_np([0] * n), bases_all[:64 * n]).any()
        holes = bases_all[:64 * n].copy()
        holes[64 * 3:64 * 5] = 0                                                # identity points among the bases
        sc = O.random_fr(970 + n, n)
        assert (gpu.best_multiexp(sc, holes) == O.best_multiexp(sc, holes, 1)).all()
    vals = [P.R - 1, P.R - 2, (P.R - 1) // 2, 1 << 253, (1 << 128) - 1, 1 << 16, (1 << 16) - 1, 1 << 15, 65537, 2, 8, 7, 9, 1 << 4, (1 << 252) + 8]
    bs = bases_all[:64 * len(vals)]
    assert (gpu.best_multiexp(fr_np(vals), bs) == O.best_multiexp(fr_np(vals), bs, 1)).all()
    assert (gpu.best_multiexp(fr_np([P.R - 1]), g) == neg).all()
    assert (gpu.best_multiexp(fr_np([5, 1]), np.concatenate([ident, g])) == g).all()
    # words that are not canonical Montgomery residues (>= r): both paths reduce them the same way
    raw = np.full(32 * 5, 0xFF, dtype=np.uint8)
    assert (gpu.best_multiexp(raw, bases_all[:64 * 5]) == pipeline(raw, bases_all[:64 * 5])).all()
    assert (gpu.best_multiexp(raw, bases_all[:64 * 5]) == gpu.best_multiexp(dev(raw), dev(bases_all[:64 * 5]))).all()


def test_fused_job_size_and_its_default(gpu):
    """a batch of commitments over a precomputed table is cut into jobs of at most 2^x entries ("msm.log_fuse_entries"; the launch
    log tells the jobs apart): the default holds the whole batch (up to 64 polynomials), a small cap cuts it, and 0 puts the
    default back -- what a caller that saved sg_get_param's answer for a parameter never set writes on the way out"""
    from circuits_halo2_amd import arithmetic as A, ffi
    k, count = 12, 20
    n = 1 << k
    g = A.g1_fixed_base_mul(_rand_fr(11, n)).cpu().numpy()
    params = gpu.ParamsKZG(k, g, g)
    params.precompute()
    cols = [_rand_fr(200 + i, n) for i in range(count)]

    def jobs():
        ffi.set_param("msm.acc_log", 1)
        try:
            pts = params.commit_batch(cols)
            return [r["M"] for r in ffi.msm_launch_log() if r["fixed"] == 1], pts
        finally:
            ffi.set_param("msm.acc_log", 0)

    try:
        whole, want = jobs()
        assert whole == [count]
        ffi.set_param("msm.log_fuse_entries", 18)      # 2^18 / (22 windows x 2^12 rows) = 2 polynomials per job
        cut, got = jobs()
        assert sum(cut) == count and max(cut) <= 3 and len(cut) >= 7
        assert all((a == b).all() for a, b in zip(got, want))
        ffi.set_param("msm.log_fuse_entries", 0)
        again, got = jobs()
        assert again == [count] and all((a == b).all() for a, b in zip(got, want))
    finally:
        ffi.set_param("msm.log_fuse_entries", 0)
        params.free()


@pytest.mark.parametrize("log_n", [1, 2, 5, 9, 12, 14, 17, 20])
def test_ntt_with_two_stages_per_sweep_writes_the_same_words(gpu, log_n):
    """"ntt.radix4" = 1 (every pass takes two DIT stages per sweep, four elements per thread: csrc/ntt_pass_body.inc) against 2
    (never): forward and inverse transforms, one- two- and three-pass plans, odd and even stage counts, batched launches and the
    coset transform with its table product on the load -- the same words (the oracle pins the default in tests/test_gpu_parity.py)"""
    import torch
    from circuits_halo2_amd import arithmetic as A, ffi
    from circuits_halo2_amd.domain import EvaluationDomain
    n = 1 << log_n
    dom = EvaluationDomain(6, log_n)
    cols = [_rand_fr(4000 + 10 * log_n + i, n) for i in range(5 if log_n <= 17 else 1)]

    def run(mode):
        ffi.set_param("ntt.radix4", mode)
        try:
            fwd = A.best_fft_batch([c.clone() for c in cols], dom.get_omega(), log_n)
            inv = A.best_fft_batch([c.clone() for c in cols], dom.get_omega_inv(), log_n, dom.ifft_divisor())
            one = dom.lagrange_to_coeff(cols[0].clone())
            cos = dom.coeff_to_cosets_batch(cols) if log_n <= 17 else []
            torch.cuda.synchronize()
            return fwd + inv + [one] + cos
        finally:
            ffi.set_param("ntt.radix4", 0)

    for a, b in zip(run(1), run(2)):
        assert (a == b).all() and a.any()
