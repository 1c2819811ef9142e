"""The entry points added to the C ABI in round 3, each against what it abbreviates or against the oracle:
sg_fr_random_batch_dev (several draws, one launch), sg_ntt_fr_batch_oop_dev (out-of-place transforms),
sg_srs_upload_dev / sg_srs_copy_dev (an SRS that arrives in device memory), the device-side setup export / import of the
batch driver, sg_collect_retired and the "lanes" parameter."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from circuits_halo2_amd import ffi
    ffi.check(ffi.lib().sg_init(0))
    return torch


def test_random_batch_equals_consecutive_single_draws():
    torch = _gpu()
    from circuits_halo2_amd import arithmetic as A, ffi
    from oracle import pyref as PR
    key = bytes(range(7, 39))
    counts = [6, 6, 1, 300, 5, 0, 1025]
    outs = [torch.zeros(32 * max(c, 1), dtype=torch.uint8, device="cuda") for c in counts]
    ptrs = (C.c_void_p * len(counts))(*[o.data_ptr() for o in outs])
    ns = (C.c_size_t * len(counts))(*counts)
    kb = np.frombuffer(key, dtype=np.uint8).copy()
    ffi.check(ffi.lib().sg_fr_random_batch_dev(ffi.ptr(kb), C.c_uint64(40), ptrs, ns, C.c_uint32(len(counts)), ffi.current_stream_ptr()))
    for d, (c, o) in enumerate(zip(counts, outs)):
        if not c:
            continue
        single = A.fr_random(key, 40 + d, c)
        assert (o[:32 * c] == single).all(), d
        want = b"".join(v.to_bytes(32, "little") for v in PR.chacha_field_elements(key, 40 + d, min(c, 40)))
        assert o[:32 * min(c, 40)].cpu().numpy().tobytes() == want
    too_many = (C.c_void_p * 9)(*[outs[0].data_ptr()] * 9)
    assert ffi.lib().sg_fr_random_batch_dev(ffi.ptr(kb), C.c_uint64(1), too_many, (C.c_size_t * 9)(*[1] * 9), C.c_uint32(9), None) == -1


@pytest.mark.parametrize("log_n", [1, 4, 9, 12, 17, 19])
def test_out_of_place_batch_transform(log_n):
    torch = _gpu()
    from circuits_halo2_amd import arithmetic as A, ffi
    from circuits_halo2_amd.domain import EvaluationDomain
    from circuits_halo2_amd.utils import random_fr_canonical
    from oracle import oracle as O
    n = 1 << log_n
    count = 5 if log_n < 19 else 2
    srcs = [A.fr_to_montgomery(torch.from_numpy(random_fr_canonical(900 + i + log_n, n)).cuda()) for i in range(count)]
    keep = [s.clone() for s in srcs]
    outs = [torch.empty_like(s) for s in srcs]
    dom = EvaluationDomain(6, log_n) if log_n >= 3 else None
    omega = np.ascontiguousarray(O.omega(log_n))
    ins_p = (C.c_void_p * count)(*[s.data_ptr() for s in srcs])
    outs_p = (C.c_void_p * count)(*[o.data_ptr() for o in outs])
    ffi.check(ffi.lib().sg_ntt_fr_batch_oop_dev(ins_p, outs_p, C.c_size_t(count), ffi.ptr(omega), None, C.c_uint32(log_n), ffi.current_stream_ptr()))
    for s, kp, o in zip(srcs, keep, outs):
        assert (s == kp).all()                                         # the inputs stay
        assert (o.cpu().numpy() == O.best_fft(kp.cpu().numpy(), omega, log_n, 4)).all()
    if dom is not None:                                                # with a divisor: lagrange_to_coeff
        w_inv, div = ffi.u8(dom.get_omega_inv()), ffi.u8(dom.ifft_divisor())
        ffi.check(ffi.lib().sg_ntt_fr_batch_oop_dev(ins_p, outs_p, C.c_size_t(count), ffi.ptr(w_inv), ffi.ptr(div), C.c_uint32(log_n), ffi.current_stream_ptr()))
        assert (outs[0].cpu().numpy() == O.lagrange_to_coeff(keep[0].cpu().numpy(), log_n, 4)).all()
    same = (C.c_void_p * 1)(srcs[0].data_ptr())
    assert ffi.lib().sg_ntt_fr_batch_oop_dev(same, same, C.c_size_t(1), ffi.ptr(omega), None, C.c_uint32(log_n), None) == -1   # aliased


def test_out_of_place_batch_refuses_vectors_that_overlap():
    """the vectors of one launch are transformed side by side, so an output that overlaps ANY input (not only its own) or
    another output would be corrupted silently: refused with SG_ERR_INVALID (advisor r03)"""
    torch = _gpu()
    from circuits_halo2_amd import ffi
    from oracle import oracle as O
    k = 8
    n = 1 << k
    buf = torch.zeros(32 * n * 4, dtype=torch.uint8, device="cuda")
    at = lambda off: buf.data_ptr() + off
    w = O.omega(k)

    def call(ins, outs):
        pi = (C.c_void_p * len(ins))(*ins)
        po = (C.c_void_p * len(outs))(*outs)
        return ffi.lib().sg_ntt_fr_batch_oop_dev(pi, po, C.c_size_t(len(ins)), ffi.ptr(ffi.u8(w)), None, C.c_uint32(k), None)
    assert call([at(0), at(32 * n)], [at(64 * n), at(96 * n)]) == 0                         # disjoint: fine
    assert call([at(0), at(32 * n)], [at(32 * n), at(96 * n)]) != 0                         # out[0] is in[1]
    assert call([at(0), at(32 * n)], [at(64 * n), at(64 * n)]) != 0                         # the same output twice
    assert call([at(0), at(32 * n)], [at(64 * n), at(64 * n + 32)]) != 0                    # outputs overlap partially
    assert call([at(0)], [at(32)]) != 0                                                    # output inside its input
    torch.cuda.synchronize()


def test_srs_from_device_memory_and_device_side_setup_import():
    torch = _gpu()
    import os
    from conftest import GOLDEN
    from circuits_halo2_amd import api, batch as B
    from circuits_halo2_amd.params import ParamsKZG
    from circuits_halo2_amd.utils import random_fr_canonical
    from circuits_halo2_amd.arithmetic import fr_to_montgomery
    host = ParamsKZG.read(open(os.path.join(GOLDEN, "hermez-raw-11"), "rb"))
    d_g, d_gl = host.device_bases()
    assert (d_g.cpu().numpy() == host.g).all() and (d_gl.cpu().numpy() == host.g_lagrange).all()
    dev = ParamsKZG.from_device(11, d_g, d_gl, host.g2, host.s_g2)
    try:
        poly = fr_to_montgomery(torch.from_numpy(random_fr_canonical(5, 1 << 11)).cuda())
        assert (dev.commit(poly) == host.commit(poly)).all() and (dev.commit_lagrange(poly) == host.commit_lagrange(poly)).all()
        dev.check()
        assert dev.write() == host.write()                              # the host copy is fetched when asked for
        with pytest.raises(ValueError):
            ParamsKZG.from_device(12, d_g, d_gl)
    finally:
        dev.free()
    # what the nccl broadcast carries: device tensors out of one key, into another, nothing through the host
    params, pk, vk = api.generate_setup_artifacts(11, os.path.join(GOLDEN, "hermez-raw-11"), api.MstInclusionCircuit.init_empty(4, 2, 8))
    try:
        setup = B.export_setup(params, pk, on_device=True)
        assert all(t.is_cuda for t in [setup["g"], setup["g_lagrange"]] + setup["fixed"] + setup["sigma"])
        assert setup["fixed"][0].data_ptr() == pk.fixed_lagrange[0].data_ptr()          # the key's own tensors, not copies
        params2, pk2, vk2 = B.import_setup(setup)
        assert vk2.fixed_comms == vk.fixed_comms and vk2.permutation_comms == vk.permutation_comms and vk2.transcript_repr == vk.transcript_repr
        assert params2._g is None and pk2.fixed_lagrange[3].data_ptr() == setup["fixed"][3].data_ptr()
        params2.free()
    finally:
        params.free()
        host.free()


def test_collect_retired_and_lane_count():
    torch = _gpu()
    import circuits_halo2_amd as sg
    from circuits_halo2_amd import ffi
    from circuits_halo2_amd.arithmetic import fr_to_montgomery, g1_fixed_base_mul
    from circuits_halo2_amd.utils import random_fr_canonical
    L = ffi.lib()
    n = 3000
    sc = fr_to_montgomery(torch.from_numpy(random_fr_canonical(71, n)).cuda())
    bases = g1_fixed_base_mul(fr_to_montgomery(torch.from_numpy(random_fr_canonical(72, n)).cuda()))
    want = sg.best_multiexp(sc, bases)
    big = fr_to_montgomery(torch.from_numpy(random_fr_canonical(73, 1 << 16)).cuda())
    sg.best_multiexp(big, g1_fixed_base_mul(big))                       # grows the work space: the small one is retired
    torch.cuda.synchronize()
    ffi.check(L.sg_collect_retired())
    assert (sg.best_multiexp(sc, bases) == want).all()
    assert L.sg_set_param(b"lanes", 0) == -1 and L.sg_set_param(b"lanes", 9) == -1
    try:
        ffi.check(L.sg_set_param(b"lanes", 2))                           # six callers, two lanes: the others wait, nobody fails
        got, errs = [], []

        def call():
            try:
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    for _ in range(5):
                        got.append(bytes(sg.best_multiexp(sc, bases)))
            except Exception as ex:      # pragma: no cover
                errs.append(ex)
        threads = [threading.Thread(target=call) for _ in range(6)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errs and len(got) == 30 and set(got) == {bytes(want)}
    finally:
        ffi.check(L.sg_set_param(b"lanes", 4))


def test_thread_binding_reports_the_bound_device():
    """sg_device / sg_bind_thread (one process per GPU: a worker thread's own HIP calls must land on the library's device);
    a tensor from another device is refused at the pointer hand-over (only checkable in full on a multi-GPU host)"""
    import torch
    from circuits_halo2_amd import ffi
    L = ffi.lib()
    ffi.check(L.sg_init(torch.cuda.current_device()))
    assert L.sg_device() == torch.cuda.current_device()
    got = []
    th = threading.Thread(target=lambda: got.append((ffi.bind_thread(), torch.cuda.current_device(), ffi.cuda_device())))
    th.start()
    th.join()
    assert got == [(L.sg_device(), L.sg_device(), torch.device("cuda", L.sg_device()))]
    t = torch.zeros(32, dtype=torch.uint8, device=ffi.cuda_device())
    assert ffi.dev_ptr(t).value == t.data_ptr()
    if torch.cuda.device_count() > 1:
        other = torch.zeros(32, dtype=torch.uint8, device=torch.device("cuda", (L.sg_device() + 1) % torch.cuda.device_count()))
        with pytest.raises(ffi.SummaGpuError):
            ffi.dev_ptr(other)


def test_calls_in_flight_are_bit_exact_under_every_sharing_setting():
    """docs/history.md 4.11: persistent accumulation (2 / 3 waves per SIMD, one ticket per wave), wave priorities, chained accumulations,
    the lean bucket reduction -- none of it may change a result bit, alone or with three calls in flight"""
    import torch
    import circuits_halo2_amd as sg
    from circuits_halo2_amd import arithmetic as A, ffi
    from circuits_halo2_amd.utils import random_fr_canonical
    _gpu()
    L = ffi.lib()
    n = 1 << 17
    scal = [A.fr_to_montgomery(torch.from_numpy(random_fr_canonical(900 + i, n)).cuda()) for i in range(3)]
    bases = A.g1_fixed_base_mul(A.fr_to_montgomery(torch.from_numpy(random_fr_canonical(77, n)).cuda()))
    torch.cuda.synchronize()
    want = [bytes(sg.best_multiexp(s, bases)) for s in scal]
    assert len(set(want)) == 3
    settings = [{"msm.acc_waves": 2, "msm.red_lean": 2}, {"msm.acc_waves": 8, "side_prio": 0, "msm.acc_chain": 0},
                {"msm.acc_waves": 3, "msm.red_lean": 0, "side_prio": 1}, {}]
    defaults = {"msm.acc_waves": 0, "msm.red_lean": 1, "side_prio": 1, "msm.acc_chain": 1}
    try:
        for setting in settings:
            for name, value in {**defaults, **setting}.items():
                ffi.check(L.sg_set_param(name.encode(), int(value)))
            got = [[] for _ in range(3)]

            def run(i):
                ffi.bind_thread()
                for _ in range(6):
                    got[i].append(bytes(sg.best_multiexp(scal[i], bases)))
            threads = [threading.Thread(target=run, args=(i,)) for i in range(3)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            assert all(g == [want[i]] * 6 for i, g in enumerate(got)), setting
    finally:
        for name, value in defaults.items():
            ffi.check(L.sg_set_param(name.encode(), int(value)))


def test_low_polynomial_alone_is_a_column():
    """sg_fr_lincomb_low_dev with no polynomials: the instance column of a proof from its few values, in one launch"""
    import torch
    from circuits_halo2_amd import ffi
    _gpu()
    n = 1 << 10
    low = np.arange(1, 6 * 8 + 1, dtype=np.uint32).view(np.uint8).copy()            # six 32-byte values (any words < r)
    out = torch.full((32 * n,), 0xAB, dtype=torch.uint8, device="cuda")
    ffi.check(ffi.lib().sg_fr_lincomb_low_dev(None, None, C.c_uint32(0), C.c_size_t(n), ffi.ptr(low), C.c_uint32(6), ffi.dev_ptr(out),
                                              ffi.current_stream_ptr()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert (got[:32 * 6] == low).all() and not got[32 * 6:].any()
    assert ffi.lib().sg_fr_lincomb_low_dev(None, None, C.c_uint32(0), C.c_size_t(n), ffi.ptr(low), C.c_uint32(9), ffi.dev_ptr(out),
                                           ffi.current_stream_ptr()) != 0
