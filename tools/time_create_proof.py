"""wall time of circuits_halo2_amd.prover.create_proof for the reference circuit in its own floor plan
(mst_inclusion.reference_assignment; a real inclusion witness from a device-built tree: LEVELS = 20 for k >= 13 -- the
reference bench's shape --, LEVELS = 4 below), k from argv (default 17); PROFILE=1 prints the host-side profile of one call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi, prover, mst_inclusion as M
from circuits_halo2_amd.utils import ints_to_fr


def setup(k):
    from full_flow import build
    asg = build(20 if k >= 13 else 4, k, user=5)
    params = sg.ParamsKZG.setup(k, ints_to_fr([0x1D0C0FFEE1234567890ABCDEF]))
    params.precompute()
    dev = lambda ints: torch.from_numpy(ints_to_fr(ints)).cuda()
    pk = prover.ProvingKey(params, k, [dev(c) for c in asg["fixed"]], [dev(c) for c in asg["sigma"]])
    fixed, sigma = [dev(c) for c in asg["fixed"]], [dev(c) for c in asg["sigma"]]
    torch.cuda.synchronize()
    t = time.perf_counter()
    pk = prover.ProvingKey(params, k, fixed, sigma)      # second construction: work spaces and plans are warm
    setup.keygen_ms = (time.perf_counter() - t) * 1e3
    return params, pk, [dev(c) for c in asg["advice"]], asg["instances"]


def run(k=17, reps=5):
    params, pk, advice, instances = setup(k)
    prover.create_proof(params, pk, advice, instances)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        proof = prover.create_proof(params, pk, advice, instances)
        best = min(best, time.perf_counter() - t)
    tm = {}
    prover.create_proof(params, pk, advice, instances, timings=tm)
    params.free()
    run.phases = {k_: round(v_, 2) for k_, v_ in tm.items()}
    run.keygen_ms = setup.keygen_ms
    return best * 1e3, len(proof)


if __name__ == "__main__":
    ffi.check(ffi.lib().sg_init(0))
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
    if os.environ.get("PROFILE"):
        import cProfile, pstats
        params, pk, advice, instances = setup(k)
        prover.create_proof(params, pk, advice, instances)
        pr = cProfile.Profile()
        pr.enable()
        prover.create_proof(params, pk, advice, instances)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
    else:
        ms, nbytes = run(k)
        print(f"create_proof k={k}: {ms:.2f} ms per proof ({nbytes} bytes), best of 5; phases (synchronised run): {run.phases}; "
              f"proving-key construction from Lagrange columns (17 commitments, 20 iNTT + 20 coset NTT): {run.keygen_ms:.2f} ms")
