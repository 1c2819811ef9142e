"""Sanitizer leg ("race detection / sanitizers"; CPU builds only -- GPU sanitizers do not exist on
this pool): the host arithmetic of the product and the C oracle rebuilt with -fsanitize=address,undefined and run over
their own checks.

* tests/cpp/host_math_check.cpp: csrc/host_curve.h, csrc/host_pairing.h, the host parts of include/summa_prover.hpp
  (Fr, Keccak, Blake2b, both transcripts, the lookup permutation) and of include/summa_circuit.hpp (floor plan, gate
  program, verifying-key digest);
* the 29-bit limb arithmetic of the kernels compiled for the host (tests/checks/limb_*_check.cpp) against Python integers;
* tools/circuit_dump.cpp (the compiled floor plan and gate programs): same bytes as the unsanitized build;
* oracle/bn254_oracle.c (`make asan`): MSM, NTT, domain operations, Poseidon tree and the quotient blocks against the
  big-integer twin, loaded into a Python child that preloads the sanitizer runtime."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
HOST = ["-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
        "-I" + os.path.join(ROOT, "circuits_halo2_amd", "csrc")]
LINK = ["-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]


def _clean(r):
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_host_arithmetic_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_math_check")
    subprocess.check_call(["g++"] + SAN + HOST + [os.path.join(ROOT, "tests", "cpp", "host_math_check.cpp"), "-o", exe] + LINK)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=ENV)
    _clean(r)
    assert "host math ok" in r.stdout


@pytest.mark.parametrize("script", ["limb_f29_check.py", "limb_curve29_check.py"])
def test_limb_arithmetic_under_asan_and_ubsan(script):
    env = dict(ENV, SG_CHECK_CXXFLAGS=" ".join(SAN))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", script)], capture_output=True, text=True, timeout=900, env=env)
    _clean(r)
    assert "bad 0" in r.stdout


def test_compiled_floor_plan_under_asan_and_ubsan(tmp_path):
    src = os.path.join(ROOT, "tools", "circuit_dump.cpp")
    outs = []
    for tag, flags in (("plain", ["-O1"]), ("san", SAN)):
        exe, out = str(tmp_path / f"circuit_dump_{tag}"), str(tmp_path / f"dump_{tag}.bin")
        subprocess.check_call(["g++"] + flags + HOST + [src, "-o", exe] + LINK)
        r = subprocess.run([exe, "11", "4", "2", "8", out], capture_output=True, text=True, timeout=600, env=ENV)
        _clean(r)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 100000


ORACLE_CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["REPO_ROOT"])
from oracle import oracle as O, pyref as P
assert "asan" in O.build()
fr = lambda xs: np.frombuffer(P.frs_to_bytes(xs), dtype=np.uint8).copy()
# MSM (threads, window rule, zero digits) and NTT against the big-integer twin
n = 257
sc = P.random_fr(11, n)
sc[3] = 0
sc[5] = P.R - 1
pts = [P.g1_mul(P.G1_GEN, s) for s in P.random_fr(12, 16)]
bases = b"".join(P.g1_to_bytes(pts[i % 16]) for i in range(n))
want = P.msm_naive(sc, [pts[i % 16] for i in range(n)])
for threads in (1, 3):
    got = O.best_multiexp(fr(sc), np.frombuffer(bases, dtype=np.uint8).copy(), threads)
    assert bytes(got) == P.g1_to_bytes(want), threads
a = P.random_fr(13, 256)
assert P.frs_from_bytes(O.best_fft(fr(a), O.omega(8), 8, 3).tobytes()) == P.ntt(a, P.omega_for(8), 8)
co = O.lagrange_to_coeff(fr(a), 8, 2)
assert P.frs_from_bytes(co.tobytes()) == P.intt(a, 8)
ext = O.coeff_to_extended(co, 8, 11, 2)
assert P.frs_from_bytes(ext.tobytes()) == P.coeff_to_extended(P.intt(a, 8), 8, 11)
back = O.extended_to_coeff(O.divide_by_vanishing_poly(ext.copy(), 8, 11), 8, 11, 2)
ext_ints = P.frs_from_bytes(ext.tobytes())
want_back = P.extended_to_coeff(P.divide_by_vanishing_poly(ext_ints, 8, 11), 8, 11)
assert P.frs_from_bytes(back.tobytes())[:len(want_back)][:5 * 256] == list(want_back)[:5 * 256]
# Poseidon tree
ents = [P.mst_entry("user%d" % i, [i + 1, 2 * i + 7]) for i in range(8)]
users, bals = fr([e[0] for e in ents]), fr([v for e in ents for v in e[1]])
h, b = O.mst_leaves(users, bals, 2), bals
_, levels = P.mst_build(ents)
for lvl in range(1, 4):
    h, b = O.mst_level(h, b, 2)
    assert P.frs_from_bytes(h.tobytes()) == [nd[0] for nd in levels[lvl]]
print("oracle under asan ok")
'''


def test_c_oracle_under_asan_and_ubsan(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    assert os.path.exists(so)
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.isabs(asan_rt), "no libasan in this toolchain"
    script = tmp_path / "child.py"
    script.write_text(ORACLE_CHILD)
    env = dict(ENV, LD_PRELOAD=asan_rt, SUMMA_ORACLE_LIB=so, REPO_ROOT=ROOT, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    _clean(r)
    assert "oracle under asan ok" in r.stdout
