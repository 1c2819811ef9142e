"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol
include/summa_gpu.h declares, argument validation and error behaviour (no GPU here => every
compute call must fail loudly, never fall back), SRS container parsing, domain bookkeeping."""
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, fr_np


def _declared_symbols(header="summa_gpu.h", prefix="sg_"):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", hdr)))


def test_abi_exports_every_declared_symbol():
    import ctypes as C
    from circuits_halo2_amd import ffi
    path = ffi.library_path()
    assert os.path.exists(path), "libsumma_gpu.so not built (run __graft_entry__.build())"
    L = C.CDLL(path)
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/summa_gpu.h but not exported"
    assert set(ffi.EXPORTS) == set(names)
    assert ffi.lib().sg_version().startswith(b"summa_gpu")
    # the compiled-host prover's ABI (include/summa_prover.h) lives in the same library
    prover_names = _declared_symbols("summa_prover.h", "sp_")
    assert set(prover_names) == set(ffi.PROVER_EXPORTS) and len(prover_names) == 6
    for n in prover_names:
        assert hasattr(L, n), f"{n} declared in include/summa_prover.h but not exported"
    assert ffi.prover_lib().sp_last_error() == b""


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import circuits_halo2_amd as sg
    assert sg.lib().sg_device_count() == 0
    with pytest.raises(sg.SummaGpuError) as e:
        sg.best_multiexp(fr_np([1]), np.zeros(64, np.uint8))
    assert e.value.code == -2  # SG_ERR_NO_DEVICE
    with pytest.raises(sg.SummaGpuError):
        sg.best_fft(fr_np([1, 2]), fr_np([1]), 1)
    with pytest.raises(sg.SummaGpuError):
        sg.EvaluationDomain(6, 4).lagrange_to_coeff(fr_np(range(16)))


def test_argument_validation_matches_upstream_assertions():
    import circuits_halo2_amd as sg
    with pytest.raises(ValueError):  # assert_eq!(coeffs.len(), bases.len())
        sg.best_multiexp(fr_np([1, 2]), np.zeros(64, np.uint8))
    with pytest.raises(ValueError):  # a.len() == 1 << log_n
        sg.best_fft(fr_np([1, 2, 3]), fr_np([1]), 2)
    with pytest.raises(ValueError):
        sg.EvaluationDomain(6, 27)  # extended_k would exceed the 2-adicity
    d = sg.EvaluationDomain(6, 11)
    assert (d.k, d.extended_k, d.quotient_poly_degree, d.extended_len()) == (11, 14, 5, 1 << 14)
    assert sg.EvaluationDomain(3, 5).extended_k == 6 and sg.EvaluationDomain(2, 5).extended_k == 5


def test_params_read_rawbytes(srs11):
    import circuits_halo2_amd as sg
    p = sg.ParamsKZG.read(open(os.path.join(GOLDEN, "hermez-raw-11"), "rb"))
    assert p.k == 11 and p.n == 2048
    assert (p.g == srs11["g_np"]).all() and (p.g_lagrange == srs11["gl_np"]).all()
    assert len(p.g2) == 128 and len(p.s_g2) == 128
    raw = open(os.path.join(GOLDEN, "hermez-raw-11"), "rb").read()
    with pytest.raises(ValueError):
        sg.ParamsKZG.read(raw[:-1])
    with pytest.raises(ValueError):
        sg.ParamsKZG.read(b"\x0b\x00")
    with pytest.raises(ValueError):
        sg.ParamsKZG(11, srs11["g_np"][:-64], srs11["gl_np"])
    with pytest.raises(ValueError):
        p.commit(fr_np(range(2049)))


def test_sharding_bookkeeping():
    from circuits_halo2_amd.batch import deal, owner_of
    from circuits_halo2_amd.distributed import shard_bounds
    # the 1024 users of a batch over 8 ranks: every proof owned exactly once, by the rank owner_of names
    users = list(range(100, 1124))
    shares = [deal(users, r, 8) for r in range(8)]
    assert sorted(u for sh in shares for u in sh) == users and all(len(sh) == 128 for sh in shares)
    assert all(owner_of(users.index(u), 8) == r for r, sh in enumerate(shares) for u in sh[:5])
    for n, w in ((1 << 20, 8), (1000, 3), (5, 8), (0, 2)):
        cover = []
        for r in range(w):
            lo, hi = shard_bounds(n, r, w)
            assert 0 <= lo <= hi <= n
            cover += list(range(lo, hi)) if n <= 1000 else []
        if n <= 1000:
            assert cover == list(range(n))
        else:
            assert shard_bounds(n, 0, w)[0] == 0 and shard_bounds(n, w - 1, w)[1] == n


def test_seeded_inputs_are_in_range():
    from circuits_halo2_amd.utils import R_MODULUS, random_fr_canonical
    c = random_fr_canonical(123, 4096)
    vals = [int.from_bytes(c[32 * i:32 * i + 32].tobytes(), "little") for i in range(4096)]
    assert max(vals) < R_MODULUS and len(set(vals)) == 4096
    assert (random_fr_canonical(123, 4096) == c).all() and not (random_fr_canonical(124, 4096) == c).all()


def test_host_point_sum_matches_bigint(srs11):
    """sg_g1_sum_affine runs on the host (no GPU needed): compare with the big-integer twin"""
    import ctypes as C
    from circuits_halo2_amd import ffi
    from oracle import pyref as P
    pts = srs11["gl_np"][:64 * 9].copy()
    pts[64 * 4:64 * 5] = 0  # an identity in the middle is skipped
    want = None
    for i in range(9):
        want = P.g1_add(want, P.g1_from_bytes(pts[64 * i:64 * i + 64].tobytes()))
    out = np.zeros(64, dtype=np.uint8)
    ffi.check(ffi.lib().sg_g1_sum_affine(ffi.ptr(pts), C.c_size_t(9), ffi.ptr(out)))
    assert P.g1_from_bytes(out.tobytes()) == want
    p = pts[:64]
    neg = np.frombuffer(P.g1_to_bytes(P.g1_neg(P.g1_from_bytes(p.tobytes()))), dtype=np.uint8)
    both = np.concatenate([p, neg, p, p])
    ffi.check(ffi.lib().sg_g1_sum_affine(ffi.ptr(both), C.c_size_t(2), ffi.ptr(out)))
    assert not out.any()
    ffi.check(ffi.lib().sg_g1_sum_affine(ffi.ptr(both[128:]), C.c_size_t(2), ffi.ptr(out)))
    assert P.g1_from_bytes(out.tobytes()) == P.g1_mul(P.g1_from_bytes(p.tobytes()), 2)
    from circuits_halo2_amd.distributed import combine_partials
    assert P.g1_from_bytes(combine_partials(pts).tobytes()) == want


def test_g2_generator_mul_matches_fixture_and_bigint(srs11):
    """sg_g2_generator_mul (host code, the verifier half of ParamsKZG::setup): 1 * G2 reproduces the g2 bytes
    of the reference's SRS container; tau * G2 and edge scalars match the big-integer group law"""
    import os
    from conftest import GOLDEN
    from circuits_halo2_amd import ffi
    from oracle import pyref as P
    raw = open(os.path.join(GOLDEN, "hermez-raw-11"), "rb").read()
    g2_fixture = raw[4 + 128 * 2048:4 + 128 * 2048 + 128]
    assert g2_fixture == P.g2_to_bytes(P.G2_GENERATOR)
    out = np.zeros(128, dtype=np.uint8)

    def mul(k):
        ffi.check(ffi.lib().sg_g2_generator_mul(ffi.ptr(np.frombuffer(P.fr_to_bytes(k % P.R), dtype=np.uint8).copy()), ffi.ptr(out)))
        return out.tobytes()

    assert mul(1) == g2_fixture
    assert mul(0) == bytes(128)
    for k in (2, 3, 0xDEADBEEF, P.R - 1, P.random_fr(5, 1)[0]):
        assert mul(k) == P.g2_to_bytes(P.g2_mul(P.G2_GENERATOR, k))
    assert mul(P.R - 1) == P.g2_to_bytes((P.G2_GENERATOR[0], ((-P.G2_GENERATOR[1][0]) % P.Q, (-P.G2_GENERATOR[1][1]) % P.Q)))


def test_header_is_plain_c_and_a_c_client_links():
    """the drop-in boundary is a C ABI: the header compiles as C99 (-pedantic) and a plain-C client links
    against the library and runs its host-side entry points (tests/c/abi_client.c)"""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    inc = os.path.join(ROOT, "include")
    for header in ("summa_gpu.h", "summa_prover.h"):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                               os.path.join(inc, header)])
    libdir = os.path.join(ROOT, "circuits_halo2_amd")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "abi_client")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", inc,
                               os.path.join(ROOT, "tests", "c", "abi_client.c"), "-o", exe, "-L", libdir, "-lsumma_gpu",
                               "-Wl,-rpath," + libdir])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
        assert "abi client ok" in out.stdout


def test_cpp_host_mirror_compiles_and_links():
    """include/summa_gpu.hpp + tests/cpp/parity_main.cpp build with g++ -Wall -Wextra -Werror (the program itself
    needs a GPU and runs under -m gpu)"""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("g++"):
        pytest.skip("no g++")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "parity_main.cpp"), "-o", os.path.join(tmp, "parity_main"),
                               "-L", os.path.join(ROOT, "circuits_halo2_amd"), "-lsumma_gpu", "-L", os.path.join(ROOT, "oracle"),
                               "-loracle"])


@pytest.mark.parametrize("k,levels,nc,nb", [(11, 4, 2, 8), (10, 3, 1, 8), (12, 5, 3, 4)])
def test_cpp_circuit_equals_the_python_twin(k, levels, nc, nb, tmp_path):
    """include/summa_circuit.hpp (the constraint system's GraphEvaluator programs, the reference circuit's floor plan, the
    witness program, the verifying-key digest, username / balance parsing as compiled host code) against
    circuits_halo2_amd/mst_inclusion.py -- byte for byte: fixed columns, sigma columns, program, both graphs"""
    import shutil
    import struct
    import subprocess
    if not shutil.which("g++") or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("no g++ / HIP headers")
    from circuits_halo2_amd import api, mst_inclusion as M, prover as P
    from circuits_halo2_amd.utils import ints_to_fr
    from oracle import pyref as PR
    exe = str(tmp_path / "circuit_dump")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "circuit_dump.cpp"), "-o", exe,
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    out = str(tmp_path / "dump.bin")
    r = subprocess.run([exe, str(k), str(levels), str(nc), str(nb), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = open(out, "rb").read()

    def graph_bytes(g):
        b = struct.pack("<I", len(g.constants)) + b"".join(g.constants)
        b += struct.pack("<I", len(g.rotations)) + struct.pack(f"<{len(g.rotations)}i", *g.rotations)
        parts, calcs = [], b""
        for cal in g.calculations:
            off, ln = 0, 0
            if len(cal) > 3:
                off, ln = len(parts), len(cal[3])
                parts.extend(cal[3])
            calcs += struct.pack("<9I", cal[0], *cal[1], *cal[2], off, ln)
        return b + struct.pack("<I", len(g.calculations)) + calcs + struct.pack("<I", len(parts)) + b"".join(struct.pack("<3I", *p) for p in parts)
    prog, n_items, n_abs, inst, rows = M.witness_program(k, levels, nc, nb)
    asg = api.MstInclusionCircuit.init_empty(levels, nc, nb).synthesize(k)
    want = struct.pack("<4I", n_items, n_abs, rows, len(inst)) + struct.pack(f"<{len(inst)}I", *inst) + prog.tobytes()
    want += b"".join(ints_to_fr(c).tobytes() for c in asg["fixed"]) + b"".join(ints_to_fr(c).tobytes() for c in asg["sigma"])
    want += graph_bytes(M.gate_graph(nc)) + graph_bytes(M.lookup_input_graph())
    groups = M.gate_challenge_exponents(nc)
    want += struct.pack("<I", len(groups)) + b"".join(struct.pack("<I%dI" % len(g), len(g), *g) for g in groups)
    rinv = pow(1 << 256, -1, PR.Q)
    pts = [((2 * i + 1) * rinv % PR.Q, (2 * i + 2) * rinv % PR.Q) for i in range(17)]
    want += P.verifying_key_digest(k, nc, pts[:11], pts[11:]).to_bytes(32, "big")
    want += ints_to_fr([int.from_bytes(PR.keccak256(b"dxGaEAii"), "big"), 11888]).tobytes()
    assert len(raw) == len(want)
    assert raw == want


@pytest.mark.parametrize("k,levels,nc", [(11, 4, 2), (13, 20, 1)])
def test_library_keygen_columns_equal_the_python_floor_plan(k, levels, nc):
    """sg_mst_inclusion_keygen_columns (host only; what api.keygen uploads) against mst_inclusion.reference_assignment of the
    empty circuit: the 11 fixed and the 6 permutation columns, row by row; bad shapes are refused"""
    import ctypes as C
    from circuits_halo2_amd import api, ffi
    from circuits_halo2_amd.utils import ints_to_fr
    n = 1 << k
    f = np.empty((11, 32 * n), dtype=np.uint8)
    s = np.empty((6, 32 * n), dtype=np.uint8)
    rows = C.c_uint32(0)
    L = ffi.lib()
    ffi.check(L.sg_mst_inclusion_keygen_columns(C.c_uint32(k), C.c_uint32(levels), C.c_uint32(nc), C.c_uint32(8), ffi.ptr(f), ffi.ptr(s),
                                                C.byref(rows)))
    asg = api.MstInclusionCircuit.init_empty(levels, nc, 8).synthesize(k)
    assert rows.value == asg["rows_used"]
    for j in range(11):
        assert (f[j] == ints_to_fr(asg["fixed"][j])).all(), j
    for j in range(6):
        assert (s[j] == ints_to_fr(asg["sigma"][j])).all(), j
    assert L.sg_mst_inclusion_keygen_columns(C.c_uint32(3), C.c_uint32(levels), C.c_uint32(nc), C.c_uint32(8), ffi.ptr(f), ffi.ptr(s), None) != 0
    assert L.sg_mst_inclusion_keygen_columns(C.c_uint32(8), C.c_uint32(20), C.c_uint32(2), C.c_uint32(8), ffi.ptr(f), ffi.ptr(s), None) != 0   # rows


def test_ahead_of_time_gate_tables_are_current():
    """circuits_halo2_amd/csrc/gates_mst_programs.inc (the reference circuit's gate programs as compile-time tables) is what
    tools/gen_gates_programs.py makes of the CURRENT lowering: a change to compile_gates or to the gate program needs the tables
    regenerated (and the library rebuilt), otherwise the ahead-of-time kernels are silently not used any more"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_gates_programs.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "gates_mst_programs.inc is stale: run tools/gen_gates_programs.py and rebuild\n" + r.stdout + r.stderr


def test_worker_threads_are_bound_to_the_library_device(monkeypatch):
    """A host thread's current HIP device starts at 0: on rank r of a multi-GPU host a worker's bare "cuda" allocations and
    streams would land on another rank's GPU.  Every pool of the package starts its threads with ffi.bind_thread; before
    the library is bound (no device here) it is a no-op that reports -1."""
    from circuits_halo2_amd import batch, ffi
    assert ffi.lib().sg_device() == -1 and ffi.bind_thread() == -1 and ffi.lib().sg_bind_thread() == 0
    seen = []
    monkeypatch.setattr(ffi, "bind_thread", lambda: seen.append(__import__("threading").current_thread().name))
    pool = batch._workers(7)                       # a size no other test uses: the pool is created here, with the patched hook
    try:
        assert list(pool.map(lambda i: i * i, range(20))) == [i * i for i in range(20)]
        assert seen and all(name.startswith("prove7") for name in seen) and len(set(seen)) == len(seen) <= 7
    finally:
        pool.shutdown()
        batch._WORKERS.pop(7, None)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("ThreadPoolExecutor(") == src.count("initializer=ffi.bind_thread")


def test_sp_key_create_refuses_programs_outside_the_constraint_system_shape():
    """include/summa_prover.h says which ConstraintSystem shapes the compiled prover is built for; what of that
    sp_key_create can see in its arguments it checks before anything reaches the device: rotations beyond -1 / 0 / +1,
    column indices beyond 11 fixed / 3 advice / 1 instance, an instance rotation, dangling intermediates.  The reference
    circuit's own programs (N_CURRENCIES 1 .. 4) pass that check (and then fail for want of a GPU on this box)."""
    import ctypes as C
    from circuits_halo2_amd import ffi, mst_inclusion as M
    from circuits_halo2_amd.arithmetic import GraphEvaluator
    L = ffi.prover_lib()

    def create(gates_graph, nc=2):
        gates, keep1 = gates_graph._struct()
        look, keep2 = M.lookup_input_graph()._struct()
        fixed, sigma = (C.c_void_p * 11)(*[8] * 11), (C.c_void_p * 6)(*[8] * 6)      # never dereferenced: the check comes first
        digest = np.zeros(32, dtype=np.uint8)
        groups = M.gate_challenge_exponents(nc)
        exps = (C.c_uint32 * sum(len(g) for g in groups))(*[e for g in groups for e in g])
        counts = (C.c_uint32 * len(groups))(*[len(g) for g in groups])
        key = C.c_uint64(0)
        rc = L.sp_key_create(C.c_uint32(6), C.c_uint64(1), fixed, sigma, ffi.ptr(digest), C.byref(gates), C.byref(look), exps, counts,
                             C.c_uint32(len(groups)), None, C.byref(key))
        return rc, L.sp_last_error().decode()

    for nc in (1, 2, 3, 4):
        rc, msg = create(M.gate_graph(nc), nc)
        assert "does not fit" not in msg, (nc, msg)               # accepted by the shape check ...
        assert rc != 0                                            # ... and no device to build the key on here

    def tampered(edit):
        g = M.gate_graph(2)
        t = GraphEvaluator()
        t.constants, t.rotations, t.calculations = list(g.constants), list(g.rotations), [tuple(c) for c in g.calculations]
        edit(t)
        return t

    def first_column_use(t, kind):
        for i, c in enumerate(t.calculations):
            for slot in (1, 2):
                if c[slot][0] == kind:
                    return i, slot
        raise AssertionError("program has no such source")

    def set_source(t, i, slot, src):
        c = list(t.calculations[i])
        c[slot] = src
        t.calculations[i] = tuple(c)

    SG_VS_INTERMEDIATE, SG_VS_FIXED, SG_VS_ADVICE = 1, 2, 3
    cases = {
        "a rotation beyond": lambda t: t.rotations.__setitem__(0, 2),
        "a column index beyond": lambda t: set_source(t, *first_column_use(t, SG_VS_ADVICE), (SG_VS_ADVICE, 3, 0)),
        "a rotation index out of range": lambda t: set_source(t, *first_column_use(t, SG_VS_FIXED), (SG_VS_FIXED, 0, len(t.rotations))),
        "not defined yet": lambda t: set_source(t, 0, 1, (SG_VS_INTERMEDIATE, 5, 0)),
    }
    for needle, edit in cases.items():
        rc, msg = create(tampered(edit))
        assert rc == -1 and "does not fit" in msg and needle in msg, (needle, rc, msg)


def test_process_wide_parameters_read_back_and_a_batch_restores_what_it_found():
    """sg_get_param / sg_abi_version and batch._ParamScope need no device: the process-wide parameters (combiner, sleeping waits) are
    plain atomics of the library.  A batch's settings are put back to what the CALLER had set when the last batch of the process
    ends, not to built-in defaults (advisor, round 4)."""
    from circuits_halo2_amd import batch as B, ffi
    assert ffi.lib().sg_abi_version() == 3
    ffi.set_param("host.wait_sleep_us", 25)
    ffi.set_param("commit.combine_wait_us", 777)
    try:
        assert ffi.get_param("host.wait_sleep_us") == 25 and ffi.get_param("commit.combine_runners") == 1
        outer = B._ParamScope({"host.wait_sleep_us": 50, "commit.combine_wait_us": 5000, "commit.combine_runners": 2})
        outer.enter()
        inner = B._ParamScope({"commit.combine_wait_us": 2000})
        inner.enter()
        assert ffi.get_param("commit.combine_wait_us") == 2000 and ffi.get_param("host.wait_sleep_us") == 50
        outer.leave()                       # the first batch ends while the second still runs: nothing is restored yet
        assert ffi.get_param("host.wait_sleep_us") == 50 and ffi.get_param("commit.combine_runners") == 2
        inner.leave()
        assert ffi.get_param("host.wait_sleep_us") == 25 and ffi.get_param("commit.combine_wait_us") == 777
        assert ffi.get_param("commit.combine_runners") == 1
    finally:
        ffi.set_param("host.wait_sleep_us", 0)
        ffi.set_param("commit.combine_wait_us", 300)
    import pytest
    with pytest.raises(ffi.SummaGpuError):
        ffi.get_param("no.such.parameter")
