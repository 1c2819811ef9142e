"""times the custom-gate interpreter at the k = 17 proof shape (ext_k = 20) on a Poseidon-flavoured program"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from circuits_halo2_amd import ffi, arithmetic as A
from circuits_halo2_amd.utils import random_fr_canonical, to_montgomery_host

ffi.check(ffi.lib().sg_init(0))
k, ext_k = 17, 20
ne = 1 << ext_k
r = lambda seed: torch.from_numpy(to_montgomery_host(random_fr_canonical(seed, ne))).cuda()
fr1 = lambda seed: to_montgomery_host(random_fr_canonical(seed, 1))
n_adv, n_fix = 3, 11
adv = [r(i) for i in range(n_adv)]; fix = [r(10 + i) for i in range(n_fix)]
vals = r(99)
b = fr1(7)
for ngates in (4, 16, 48):
    g = A.GraphEvaluator()
    gates = []
    for t in range(ngates):
        # q_t * ( m0 * (a0 + rc)^5 + m1 * (a1 + rc')^5 - a_t(next) )   -- the shape of a Poseidon full round
        terms = []
        for j in range(2):
            x = g.add_calculation(A.ADD, g.query(A.ADVICE, j, 0), g.add_constant(fr1(100 + 2 * t + j)))
            x2 = g.add_calculation(A.SQUARE, x)
            x4 = g.add_calculation(A.SQUARE, x2)
            x5 = g.add_calculation(A.MUL, x4, x)
            terms.append(g.add_calculation(A.MUL, x5, g.add_constant(fr1(300 + 2 * t + j))))
        s = g.add_calculation(A.ADD, terms[0], terms[1])
        d = g.add_calculation(A.SUB, s, g.query(A.ADVICE, t % n_adv, 1))
        gates.append(g.add_calculation(A.MUL, g.query(A.FIXED, t % n_fix, 0), d))
    g.add_calculation(A.HORNER, (A.PREVIOUS_VALUE, 0, 0), (A.Y, 0, 0), gates)
    fn = lambda: A.quotient_gates(vals, g, fix, adv, [], np.zeros(0, dtype=np.uint8), b, b, b, b, k, ext_k)
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    nprod = sum(1 for c in g.calculations if c[0] in (A.MUL, A.SQUARE)) + ngates  # + Horner products
    print(f"{ngates} gates, {len(g.calculations)} calculations, ~{nprod} products/row: {ms:.3f} ms per pass, "
          f"{nprod * ne / ms / 1e6:.1f} G products/s")
