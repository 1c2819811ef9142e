"""repeated proofs from both drivers (pooled buffers, side streams, device RNG), every Python-driver proof and the last C++ one verified by the restated verifier; run by hand on the GPU box"""
import sys, os, json, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for sub in ('', 'tests', 'tools'):
    sys.path.insert(0, os.path.join(ROOT, sub))
import torch
import circuits_halo2_amd as sg
from circuits_halo2_amd import ffi, prover
from circuits_halo2_amd.utils import ints_to_fr
from oracle import pyref as PR, summa_verifier as SV
import full_flow
ffi.check(ffi.lib().sg_init(0))
for k, levels in ((11, 4), (13, 20)):
    asg = full_flow.build(levels, k, user=3)
    params = sg.ParamsKZG.setup(k, ints_to_fr([0x77777]))
    pk, advice, proof = full_flow.keygen_and_prove(asg, k, params, reps=1)
    f2 = lambda b: (PR.fq_from_bytes(b[:32]), PR.fq_from_bytes(b[32:64]))
    s_g2 = (f2(params.s_g2[:64]), f2(params.s_g2[64:]))
    vk = {"k": k, "vk_digest": pk.vk_digest, "fixed_comms": pk.fixed_comms, "permutation_comms": pk.permutation_comms,
          "g2": PR.G2_GENERATOR, "neg_s_g2": (s_g2[0], ((-s_g2[1][0]) % PR.Q, (-s_g2[1][1]) % PR.Q))}
    # many proofs from the Python driver, each verified
    ok = 0
    for i in range(40):
        p = prover.create_proof(params, pk, advice, asg["instances"])
        ok += SV.verify(p, asg["instances"], vk)
    print(f"k={k}: python driver {ok}/40 proofs verified", flush=True)
    with tempfile.TemporaryDirectory() as td:
        prover.export_bundle(os.path.join(td, "b.bin"), params, pk, advice, asg["instances"])
        r = subprocess.run([os.path.join(ROOT, 'tools', 'create_proof_cpp'), os.path.join(td, "b.bin"), os.path.join(td, "p.bin"), "400"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        print(r.stdout.strip()[:90])
        print(f"k={k}: c++ driver, proof after 400 runs verifies:", SV.verify(open(os.path.join(td, "p.bin"), "rb").read(), asg["instances"], vk), flush=True)
    params.free()
