"""evaluate_h's custom-gate block with the reference circuit's constraint system (circuits_halo2_amd.mst_inclusion):
time per call on the extended domain of k = 11 .. 17 (ext_k = k + 3), random columns"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi, arithmetic as A, mst_inclusion as M
from circuits_halo2_amd.utils import random_fr_canonical
from circuits_halo2_amd.arithmetic import fr_to_montgomery

ffi.check(ffi.lib().sg_init(0))
g = M.gate_graph()
print(f"program: {len(g.calculations)} calculations, {len(g.constants)} constants, rotations {g.rotations}, "
      f"lowered: {A.gates_program_info(g, M.NUM_FIXED, M.NUM_ADVICE, 0, len(M.gate_challenge_exponents()))} (instructions, LDS slots)", flush=True)
for k in [int(x) for x in os.environ.get("KS", "11,13,15,17").split(",")]:
    ext_k = k + 3
    ne = 1 << ext_k
    col = lambda s: fr_to_montgomery(torch.from_numpy(random_fr_canonical(s, ne)).cuda())
    fixed = [col(100 + i) for i in range(M.NUM_FIXED)]
    advice = [col(200 + i) for i in range(M.NUM_ADVICE)]
    values = col(300)
    b = np.frombuffer(bytes(range(1, 33)), dtype=np.uint8).copy(); b[31] = 0
    none = np.zeros(0, dtype=np.uint8)
    chal = M.gate_challenges(0x1234567)
    best = 1e9
    for _ in range(6):
        torch.cuda.synchronize(); t = time.perf_counter()
        A.quotient_gates(values, g, fixed, advice, [], chal, b, b, b, b, k, ext_k)   # the challenges: powers of y
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"k={k} (2^{ext_k} rows, 14 columns): {best * 1e3:.3f} ms, {ne / best / 1e9:.2f} G rows/s, "
          f"{14 * 32 * ne / best / 1e9:.0f} GB/s of column reads", flush=True)
