"""A loop of lone k = 17 proofs from the compiled driver (through sp_create_proof), bracketed by torch.cuda.synchronize(), for
`rocprofv3 --hip-trace` + tools/hip_api_counts.py --window: the HIP API calls of ONE proof and the host time inside them.
usage (GPU box): rocprofv3 --hip-trace -d OUT -- python3 tools/lone_proof_api_trace.py [reps=30]; python tools/hip_api_counts.py OUT <reps> --window"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from circuits_halo2_amd import ffi, prover
from time_create_proof import setup

ffi.check(ffi.lib().sg_init(0))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
params, pk, advice, instances = setup(17)
for _ in range(5):
    prover.create_proof_native(params, pk, advice, instances, "evm", sanity_checks=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    prover.create_proof_native(params, pk, advice, instances, "evm", sanity_checks=False)
dt = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"{reps} proofs, {dt / reps * 1e3:.3f} ms each (advice columns cloned per proof)")
