"""The 9 x 29-bit limb arithmetic (csrc/bn254_f29.cuh, bn254_curve29.cuh) also compiles for
the host; these drivers check it against Python integers / the big-integer twin: products
with lazy operand bounds, lazy add/sub constants, canonicalisation, the zero-mod-p filter,
format conversions, and every branch of the XYZZ point formulas."""
import os
import subprocess
import sys

from conftest import ROOT


def _run(script):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "bad 0" in r.stdout


def test_field_limbs_vs_bigints():
    _run("limb_f29_check.py")


def test_curve_limbs_vs_bigints():
    _run("limb_curve29_check.py")
