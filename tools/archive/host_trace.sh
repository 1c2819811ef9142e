# the host-side timeline of the compiled prover (SG_PROVER_TRACE), k = 17: where the driver is when, without a profiler attached
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
SG_PROVER_TRACE=1 ./tools/create_proof_cpp gpurun_out/r02g/bundle17.bin gpurun_out/r02g/proof.bin 4 2> gpurun_out/r02g/host_trace.txt
tail -30 gpurun_out/r02g/host_trace.txt
rm -f gpurun_out/r02g/bundle17.bin
