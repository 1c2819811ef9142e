# A/B: side streams of the compiled prover at normal vs lowest priority (k = 17, best of 30 each, alternating)
set -e
mkdir -p gpurun_out/r02g
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
prover.export_bundle("gpurun_out/r02g/bundle17.bin", params, pk, adv, c.instances()[0])
PY
for i in 1 2 3; do
  for p in 0 1; do
    echo -n "SG_SIDE_PRIORITY=$p: "; SG_SIDE_PRIORITY=$p ./tools/create_proof_cpp gpurun_out/r02g/bundle17.bin gpurun_out/r02g/proof.bin 30 | cut -c1-60
  done
done
rm -f gpurun_out/r02g/bundle17.bin
