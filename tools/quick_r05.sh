#!/bin/bash
# quick look at one k = 17 proof of the compiled driver: wall clock (real and serial schedule) and the serial kernel listing
# usage (GPU box): TAG=r05b tools/quick_r05.sh
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag="${TAG:-r05b}"
work="$GRAFT_REPO_ROOT/gpurun_out/${tag}_work"
mkdir -p "$work"
python - "$work/bundle17.bin" <<'PY'
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
prover.export_bundle(sys.argv[1], params, pk, adv, c.instances()[0])
print("bundle written")
PY
CP="$GRAFT_REPO_ROOT/tools/create_proof_cpp"
"$CP" "$work/bundle17.bin" "$work/proof.bin" 30 | tee "gpurun_out/${tag}_create_proof_cpp.json"
SG_PROVER_SERIAL=1 "$CP" "$work/bundle17.bin" "$work/proof.bin" 30 | tee "gpurun_out/${tag}_create_proof_cpp_serial.json"
if [ -n "${AB_PARAMS:-}" ]; then   # e.g. AB_PARAMS="quotient.fused_numerator=0 ntt.coset_scale_pass=1"
  for p in $AB_PARAMS; do
    echo "# SG_PARAMS=$p"
    SG_PARAMS="$p" "$CP" "$work/bundle17.bin" "$work/proof.bin" 30
  done
fi
rm -rf "$work/serial_trace"
(cd /tmp && SG_PROVER_SERIAL=1 rocprofv3 --kernel-trace --stats -d "$work/serial_trace" -- "$CP" "$work/bundle17.bin" "$work/proof.bin" 8 > "$work/cpp_trace.json" 2> "$work/rocprof_trace.err")
python tools/proof_kernels.py "$work/serial_trace" | tee "gpurun_out/${tag}_proof_kernels_serial.txt"
rm -rf "$work"
