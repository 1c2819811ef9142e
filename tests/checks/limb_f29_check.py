#!/usr/bin/env python3
"""Checks bn254_f29.cuh (compiled for the host) against Python integers."""
import os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as P
M = (1 << 29) - 1
exe = "/tmp/limb_f29_check"
# SG_CHECK_CXXFLAGS: other compiler flags for the driver (tests/test_sanitizers_cpu.py: -fsanitize=address,undefined)
flags = os.environ.get("SG_CHECK_CXXFLAGS", "-O2").split()
if flags != ["-O2"]:
    exe += "_flagged"
subprocess.check_call(["g++"] + flags + ["-std=c++17", "-o", exe, os.path.join(ROOT, "tests/checks/limb_f29_check.cpp")])
def limbs(x): return [(x >> (29 * i)) & M if i < 8 else x >> (29 * i) for i in range(9)]
def val(l): return sum(v << (29 * i) for i, v in enumerate(l))
def lazy(x, rnd):
    """a non-canonical limb vector with the same value: limbs < 2^29 + 4"""
    l = limbs(x)
    for i in range(8):
        d = rnd.randrange(0, 4)
        if l[i + 1] >= 1 and l[i] + (1 << 29) * 0 + d * 0 >= 0:
            pass
    return l
def lazy_limbs(x, rnd):
    """the same value with limbs up to 2^29 + 3 where the limb above can lend (what a carry step leaves)"""
    l = limbs(x)
    for i in range(8):
        d = rnd.randrange(0, 4)
        if l[i + 1] >= 1 and l[i] < d and rnd.random() < 0.5:   # borrow 2^29 from the limb above: l[i] + 2^29 <= 2^29 + 3
            l[i] += 1 << 29
            l[i + 1] -= 1
    return l
rnd = random.Random(1)
cases = []
for fld, p in (("q", P.Q), ("r", P.R)):
    R = 1 << 261
    for _ in range(300):
        ba, bb = rnd.choice([(1, 1), (2, 8), (13, 13), (32, 2), (10, 10), (170, 1)])
        a, b = rnd.randrange(ba * p), rnd.randrange(bb * p)
        if rnd.random() < 0.1: a = rnd.choice([0, p, 2 * p, p - 1, ba * p - 1])
        cases.append(("mul", fld, limbs(a), limbs(b), ("mul", p, a, b)))
        cases.append(("add", fld, limbs(a), limbs(b), ("add", p, a, b)))
        if ba * ba <= 170:
            cases.append(("sqr", fld, limbs(a), limbs(0), ("mul", p, a, a)))
        if 2 * ba * bb <= 170:
            cases.append(("mul2", fld, limbs(a), limbs(b), ("mul2", p, a, b)))
        cases.append(("muladd", fld, limbs(a), limbs(b), ("muladd", p, a, b, ba * bb, bb)))
        if ba * bb + bb * bb + ba * ba <= 170:
            cases.append(("dot3", fld, limbs(a), limbs(b), ("dot", p, a * b + b * b + a * a)))
        if 3 * ba * bb + bb * bb + ba * ba <= 170:
            cases.append(("dot5", fld, limbs(a), limbs(b), ("dot", p, 3 * a * b + b * b + a * a)))
        # the lazily carried sums of the point additions: operands as products leave them (exactly normalised; rr < 2p; ppp and qq are
        # products of bounds <= 20, i.e. below (20 / 170.7 + 1) p < 9p/8 -- what keeps rr + 6p - ppp - 2qq above p and its top limb true)
        pa = rnd.choice([0, 1, p - 1, p, 2 * p - 1]) if rnd.random() < 0.3 else rnd.randrange(2 * p)
        pb = rnd.choice([0, 1, p - 1, p, 9 * p // 8 - 1]) if rnd.random() < 0.3 else rnd.randrange(9 * p // 8)
        cases.append(("x3nc", fld, limbs(pa), limbs(pb), ("x3nc", p, pa, pb)))
        x8 = rnd.choice([0, 29 * p // 4 - 1, p]) if rnd.random() < 0.2 else rnd.randrange(29 * p // 4)   # x3 < rr + 6p < 7.25 p
        cases.append(("tnc", fld, limbs(pa), lazy_limbs(x8, rnd), ("tnc", p, pa, x8)))
        a8, b8 = rnd.randrange(8 * p), rnd.randrange(8 * p)
        cases.append(("sub8", fld, limbs(a8), limbs(b8), ("sub", p, a8, b8, 8)))
        b2 = rnd.randrange(2 * p)
        cases.append(("sub2", fld, limbs(a8), limbs(b2), ("sub", p, a8, b2, 2)))
        b64 = rnd.randrange(64 * p)
        cases.append(("sub64", fld, limbs(a8), limbs(b64), ("sub", p, a8, b64, 64)))
        cases.append(("canon", fld, limbs(a), limbs(0), ("canon", p, a)))
        st = rnd.choice([0, p, 170 * p - 1, p - 1]) if rnd.random() < 0.1 else rnd.randrange(rnd.choice([1, 2, 25, 170]) * p)
        cases.append(("step", fld, limbs(st), limbs(0), ("step", p, st)))
        rs = rnd.choice([0, p - 1, p, p + 1, 2 * p - 1, 2 * p, 169 * p, 170 * p - 1, 64 * p - 1, 64 * p]) if rnd.random() < 0.3 else rnd.randrange(rnd.choice([1, 2, 3, 32, 170]) * p)
        cases.append(("redsmall", fld, lazy_limbs(rs, rnd), limbs(0), ("redsmall", p, rs)))
        ai = rnd.choice([0, 1, 2, p - 1, p, p + 1, 3 * p, (1 << 261) % p]) if rnd.random() < 0.2 else rnd.randrange(rnd.choice([1, 2, 64, 170]) * p)
        cases.append(("inv", fld, limbs(ai), limbs(0), ("inv", p, ai)))
        z = rnd.choice([k * p for k in range(0, 65)] + [rnd.randrange(64 * p) for _ in range(8)] + [k * p + 1 for k in range(3)])
        cases.append(("iszero", fld, limbs(z), limbs(0), ("iszero", p, z)))
        x = rnd.randrange(p)
        w = [(x >> (32 * i)) & 0xffffffff for i in range(8)] + [0]
        cases.append(("from0", fld, w, limbs(0), ("from", x, 0)))
        cases.append(("from5", fld, w, limbs(0), ("from", x, 5)))
        cases.append(("towords", fld, limbs(x), limbs(0), ("towords", x)))
inp = "\n".join(f"{op} {f} " + " ".join("%x" % v for v in a + b) for op, f, a, b, _ in cases) + "\n"
out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.strip().split("\n")
assert len(out) == len(cases)
bad = 0
for (op, f, a, b, exp), line in zip(cases, out):
    l = [int(v, 16) for v in line.split()]
    ok = True
    if exp[0] == "mul":
        _, p, x, y = exp
        v = val(l)
        ok = v % p == (x * y * pow(1 << 261, -1, p)) % p and v < 2 * p and all(t < (1 << 29) for t in l[:8])
    elif exp[0] == "mul2":
        _, p, x, y = exp
        v = val(l)
        ok = v % p == (2 * x * y * pow(1 << 261, -1, p)) % p and v < 2 * p and all(t < (1 << 29) for t in l[:8])
    elif exp[0] == "muladd":   # x y 2^-261 + y: the EXACT integer (x y + m p) / 2^261 + y, limbs 0..7 exactly normalised
        _, p, x, y, bxy, bz = exp
        v = val(l)
        ok = (v - y) % p == (x * y * pow(1 << 261, -1, p)) % p and y <= v < 2 * p + y and all(t < (1 << 29) for t in l[:8])
    elif exp[0] == "dot":
        _, p, sxy = exp
        v = val(l)
        ok = v % p == (sxy * pow(1 << 261, -1, p)) % p and v < 2 * p and all(t < (1 << 29) for t in l[:8])
    elif exp[0] == "x3nc":   # the exact integer, limbs 0..7 below 2^29 + 8 after the one carry step, top limb the true one
        _, p, x, y = exp
        ok = val(l) == x + 6 * p - 3 * y and all(t < (1 << 29) + 8 for t in l[:8]) and l[8] < (1 << 31)
    elif exp[0] == "tnc":    # the exact integer; limbs 0..6 uncarried but below 2^31, limb 7 below 2^29, top limb the true one
        _, p, x, y = exp
        ok = val(l) == x + 8 * p - y and all(t < (1 << 31) for t in l[:7]) and l[7] < (1 << 29) and l[8] < (1 << 31)
    elif exp[0] == "add":
        _, p, x, y = exp
        ok = val(l) == x + y and all(t < (1 << 29) + 4 for t in l[:8])
    elif exp[0] == "sub":
        _, p, x, y, k = exp
        ok = val(l) == x - y + k * p and all(t < (1 << 29) + 4 for t in l[:8])
    elif exp[0] == "canon":
        _, p, x = exp
        ok = val(l) == x % p and all(t < (1 << 29) for t in l[:8])
    elif exp[0] == "step":    # one Montgomery limb step: x * 2^-29 mod p, < 2p, exactly normalised limbs
        _, p, x = exp
        v = val(l)
        ok = v % p == x * pow(1 << 29, -1, p) % p and v < 2 * p and all(t < (1 << 29) for t in l[:8])
    elif exp[0] == "redsmall":   # same residue, < 2p, exactly normalised, no change of domain
        _, p, x = exp
        v = val(l)
        ok = v % p == x % p and v < 2 * p and all(t < (1 << 29) for t in l[:8])
    elif exp[0] == "inv":     # (x^)^-1 in the 2^261 domain: A -> A^-1 * 2^522 (0 -> 0), canonical
        _, p, x = exp
        want = 0 if x % p == 0 else pow(x, -1, p) * pow(2, 522, p) % p
        ok = val(l) == want and all(t < (1 << 29) for t in l[:8])
    elif exp[0] == "iszero":
        _, p, z = exp
        ok = l[0] == (1 if z % p == 0 else 0)
    elif exp[0] == "from":
        _, x, sh = exp
        ok = val(l) == x << sh and all(t < (1 << 29) for t in l)
    elif exp[0] == "towords":
        ok = sum(v << (32 * i) for i, v in enumerate(l[:8])) == exp[1]
    if not ok:
        bad += 1
        if bad < 10: print("FAIL", op, f, exp[:1], line)
print("cases", len(cases), "bad", bad)
sys.exit(1 if bad else 0)
