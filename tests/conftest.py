import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
collect_ignore_glob = ["checks/*"]   # stand-alone checkers (driven by test_limb29_cpu.py / run by hand on the GPU box)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    return json.load(open(os.path.join(GOLDEN, "kat.json")))


@pytest.fixture(scope="session")
def srs11():
    """The reference's SRS fixture backend/ptau/hermez-raw-11 (byte copy), parsed."""
    from oracle import pyref as P
    raw = open(os.path.join(GOLDEN, "hermez-raw-11"), "rb").read()
    s = P.parse_srs(raw)
    s["g_np"] = np.frombuffer(s["g"], dtype=np.uint8).copy()
    s["gl_np"] = np.frombuffer(s["g_lagrange"], dtype=np.uint8).copy()
    return s


def golden_bin(name):
    return np.fromfile(os.path.join(GOLDEN, name), dtype=np.uint8)


def fr_np(values):
    """list of ints -> Montgomery 32-B little-endian numpy buffer"""
    from oracle import pyref as P
    return np.frombuffer(P.frs_to_bytes(values), dtype=np.uint8).copy()


def point_np(xy):
    from oracle import pyref as P
    return np.frombuffer(P.g1_to_bytes(xy), dtype=np.uint8).copy()
