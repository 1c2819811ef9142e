#!/bin/bash
# the compiled prover's own host timeline (SG_PROVER_TRACE) of the last of a few proofs at k = 17
set -euo pipefail
mkdir -p gpurun_out/r03g
[ -f gpurun_out/r03g/bundle17.bin ] || python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from bench import snapshot_tree
from circuits_halo2_amd import api, ffi, prover
ffi.check(ffi.lib().sg_init(0))
tree = snapshot_tree(20, 2)
params, pk, vk = api.generate_setup_artifacts(17, None, api.MstInclusionCircuit.init_empty(20, 2, 8))
c = api.MstInclusionCircuit.init_from_tree(tree, 5)
adv = api._advice_columns(pk, c)
prover.export_bundle("gpurun_out/r03g/bundle17.bin", params, pk, adv, c.instances()[0])
PY
SG_PROVER_TRACE=1 ./tools/create_proof_cpp gpurun_out/r03g/bundle17.bin gpurun_out/r03g/proof.bin 6 > gpurun_out/proof_host_trace.json 2> gpurun_out/proof_host_trace.err
grep -n "us (+" gpurun_out/proof_host_trace.err | tail -45
cat gpurun_out/proof_host_trace.json | head -c 400
