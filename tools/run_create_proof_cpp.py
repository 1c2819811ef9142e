"""exports a bundle of the time_create_proof.setup(k) witness (reference floor plan) at k (default 17) and runs tools/create_proof_cpp on it"""
import os, sys, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from circuits_halo2_amd import ffi, prover
from time_create_proof import setup

ffi.check(ffi.lib().sg_init(0))
k = int(sys.argv[1]) if len(sys.argv) > 1 else 17
params, pk, advice, instances = setup(k)
d = tempfile.mkdtemp()
prover.export_bundle(os.path.join(d, "bundle.bin"), params, pk, advice, instances)
params.free()
ffi.lib().sg_shutdown()
torch.cuda.empty_cache()
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "create_proof_cpp")
r = subprocess.run([exe, os.path.join(d, "bundle.bin"), os.path.join(d, "proof.bin"), sys.argv[2] if len(sys.argv) > 2 else "8"],
                   capture_output=True, text=True, env=dict(os.environ, SG_PROVER_VERBOSE="1"))
print(r.stderr[-1500:])
print(r.stdout.strip())
sys.exit(r.returncode)
