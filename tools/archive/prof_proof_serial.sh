# kernel trace of the compiled-host prover at k = 17 with SG_PROVER_SERIAL=1: every kernel of a proof alone on the GPU (isolated durations)
set -e
mkdir -p gpurun_out/r02g
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from bench import snapshot_tree
from circuits_halo2_amd import api, ffi, prover
ffi.check(ffi.lib().sg_init(0))
tree = snapshot_tree(20, 2)
params, pk, vk = api.generate_setup_artifacts(17, None, api.MstInclusionCircuit.init_empty(20, 2, 8))
c = api.MstInclusionCircuit.init_from_tree(tree, 5)
adv = api._advice_columns(pk, c)
prover.export_bundle("gpurun_out/r02g/bundle17.bin", params, pk, adv, c.instances()[0])
print("bundle written")
PY
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (relative paths below are removed and written under the repo copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
# (HIP default number of hardware queues)
rm -rf gpurun_out/prof_serial
SG_PROVER_SERIAL=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -- ./tools/create_proof_cpp gpurun_out/r02g/bundle17.bin gpurun_out/r02g/proof.bin 8 > gpurun_out/r02g/cpp.json 2> gpurun_out/r02g/rocprof.err
cat gpurun_out/r02g/cpp.json
rm -f gpurun_out/r02g/bundle17.bin
python tools/proof_kernels.py gpurun_out/prof_serial > gpurun_out/serial_proof_kernels.txt
