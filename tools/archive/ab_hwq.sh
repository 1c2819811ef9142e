#!/bin/bash
# hardware queues (GPU_MAX_HW_QUEUES; HIP's default is 4, streams are assigned round-robin): single proof, batch of proofs
set -euo pipefail
mkdir -p gpurun_out/r03g
python - <<'PY'
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
for rep in 1 2; do
for q in 4 8 16; do
  echo -n "GPU_MAX_HW_QUEUES=$q proof "; GPU_MAX_HW_QUEUES=$q ./tools/create_proof_cpp gpurun_out/r03g/bundle17.bin gpurun_out/r03g/proof.bin 30 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['create_proof_ms'])"
done
done
rm -f gpurun_out/r03g/bundle17.bin
for q in 4 8 16 4 8 16; do echo -n "GPU_MAX_HW_QUEUES=$q  "; GPU_MAX_HW_QUEUES=$q python tools/run_batch.py 16 768 1 2>&1 | tail -1; done
for q in 4 8 16; do echo -n "GPU_MAX_HW_QUEUES=$q  "; GPU_MAX_HW_QUEUES=$q python tools/run_batch.py 8 768 1 2>&1 | tail -1; done
