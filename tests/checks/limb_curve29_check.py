#!/usr/bin/env python3
import os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as P
exe = "/tmp/limb_curve29_check"
# SG_CHECK_CXXFLAGS: other compiler flags for the driver (tests/test_sanitizers_cpu.py: -fsanitize=address,undefined)
flags = os.environ.get("SG_CHECK_CXXFLAGS", "-O2").split()
if flags != ["-O2"]:
    exe += "_flagged"
subprocess.check_call(["g++"] + flags + ["-std=c++17", "-o", exe, os.path.join(ROOT, "tests/checks/limb_curve29_check.cpp")])
rnd = random.Random(7)
def words(pt):
    b = P.g1_to_bytes(pt)
    return " ".join("%x" % int.from_bytes(b[4*i:4*i+4], "little") for i in range(16))
def decode(line):
    w = [int(x, 16) for x in line.split()]
    vals = []
    for c in range(4):
        v = sum(w[8*c+i] << (32*i) for i in range(8))
        vals.append(v * pow(1 << 256, -1, P.Q) % P.Q)
    X, Y, ZZ, ZZZ = vals
    if ZZ == 0: return None
    return (X * pow(ZZ, -1, P.Q) % P.Q, Y * pow(ZZZ, -1, P.Q) % P.Q)
pts = [P.g1_mul(P.G1_GEN, rnd.randrange(1, P.R)) for _ in range(40)]
cmds, expect = [], []
acc = None; other = None
def emit(c): cmds.append(c)
def dump():
    emit("dump"); expect.append(acc)
# plain chain with negations
for i, p in enumerate(pts[:20]):
    neg = i % 3 == 0
    emit(f"madd {words(p)} {int(neg)}"); acc = P.g1_add(acc, P.g1_neg(p) if neg else p); dump()
# doubling via madd of the same point onto a single-point accumulator, and cancellation
emit("reset"); acc = None
emit(f"madd {words(pts[5])} 0"); acc = pts[5]
emit(f"madd {words(pts[5])} 0"); acc = P.g1_add(acc, pts[5]); dump()
emit(f"madd {words(pts[5])} 0"); acc = P.g1_add(acc, pts[5]); dump()
emit("reset"); acc = None
emit(f"madd {words(pts[6])} 0"); emit(f"madd {words(pts[6])} 1"); acc = None; dump()
emit(f"madd {words(None)} 0"); dump()
emit(f"madd {words(pts[7])} 1"); acc = P.g1_neg(pts[7]); dump()
# accumulate A, swap, accumulate B, add
emit("reset"); acc = None
for p in pts[20:30]: emit(f"madd {words(p)} 0"); acc = P.g1_add(acc, p)
emit("swap"); other, acc = acc, other
acc = None; emit("reset")
for p in pts[30:40]: emit(f"madd {words(p)} 0"); acc = P.g1_add(acc, p)
emit("addother"); acc = P.g1_add(acc, other); dump()
emit("double"); acc = P.g1_add(acc, acc); dump()
emit("double"); acc = P.g1_add(acc, acc); dump()
# add equal accumulators (doubling branch of add) and opposite
emit("swap"); other, acc = acc, other   # other = S
emit("reset"); acc = None
for p in pts[:4]: emit(f"madd {words(p)} 0"); acc = P.g1_add(acc, p)
emit("swap"); other, acc = acc, other   # other = T, acc = S (discard)
emit("reset"); acc = None
for p in pts[:4][::-1]: emit(f"madd {words(p)} 0"); acc = P.g1_add(acc, p)   # same sum, different Z
emit("addother"); acc = P.g1_add(acc, other); dump()
emit("swap"); other, acc = acc, other
emit("reset"); acc = None
for p in pts[:4]: emit(f"madd {words(p)} 1"); acc = P.g1_add(acc, P.g1_neg(p))
emit("double"); acc = P.g1_add(acc, acc)
emit("addother"); acc = P.g1_add(acc, other); dump()     # 2T + (-2T)... other = 2T: identity
# long chain to exercise bounds
emit("reset"); acc = None
for i in range(300):
    p = pts[rnd.randrange(40)]; neg = rnd.random() < 0.5
    emit(f"madd {words(p)} {int(neg)}"); acc = P.g1_add(acc, P.g1_neg(p) if neg else p)
    if i % 50 == 49: dump()
out = subprocess.run([exe], input="\n".join(cmds) + "\n", capture_output=True, text=True, check=True)
lines = out.stdout.strip().split("\n")
assert len(lines) == len(expect), (len(lines), len(expect))
bad = sum(1 for l, e in zip(lines, expect) if decode(l) != e)
tops = [tuple(int(x, 16) for x in l.split()[2:]) for l in out.stderr.strip().split("\n")]
print("dumps", len(lines), "bad", bad, "max top limbs", [hex(max(t[i] for t in tops)) for i in range(4)], "p top", hex(0x30644e))
sys.exit(1 if bad else 0)
