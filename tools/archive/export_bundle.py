"""writes the prover bundle of the time_create_proof.setup(k) witness (reference floor plan) at k (argv[1], default 17) to argv[2] (for tools/create_proof_cpp)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from circuits_halo2_amd import ffi, prover
from time_create_proof import setup

ffi.check(ffi.lib().sg_init(0))
params, pk, advice, instances = setup(int(sys.argv[1]) if len(sys.argv) > 1 else 17)
prover.export_bundle(sys.argv[2], params, pk, advice, instances)
params.free()
