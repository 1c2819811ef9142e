#!/bin/bash
# Kernel traces of the compiled-host prover at k = 17 (tools/create_proof_cpp):
#   serial   SG_PROVER_SERIAL=1: every kernel of a proof alone on the GPU (isolated durations, their sum)
#   overlap  the real schedule (main stream, side streams, commitment jobs): the timeline of one proof
# usage (on the GPU box, from the repo root): tools/prof_proof_r04.sh <tag>     -> gpurun_out/<tag>_*.txt
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
tag="${1:-r04}"
cd "$GRAFT_REPO_ROOT"
out="$GRAFT_REPO_ROOT/gpurun_out"
work="$out/${tag}_work"
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
export TMPDIR=/tmp
rm -rf "$work/prof_serial" "$work/prof_overlap"
(cd /tmp && SG_PROVER_SERIAL=1 rocprofv3 --kernel-trace --stats -d "$work/prof_serial" -- "$GRAFT_REPO_ROOT/tools/create_proof_cpp" "$work/bundle17.bin" "$work/proof.bin" 8 > "$work/cpp_serial.json" 2> "$work/rocprof_serial.err")
cat "$work/cpp_serial.json"
python tools/proof_kernels.py "$work/prof_serial" > "$out/${tag}_proof_kernels_serial.txt"
python tools/proof_timeline_dump.py "$work/prof_serial" > "$out/${tag}_proof_timeline_serial.txt"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$work/prof_overlap" -- "$GRAFT_REPO_ROOT/tools/create_proof_cpp" "$work/bundle17.bin" "$work/proof.bin" 8 > "$work/cpp_overlap.json" 2> "$work/rocprof_overlap.err")
cat "$work/cpp_overlap.json"
python tools/proof_kernels.py "$work/prof_overlap" > "$out/${tag}_proof_kernels.txt"
python tools/proof_timeline_dump.py "$work/prof_overlap" > "$out/${tag}_proof_timeline.txt"
# without the profiler: the wall clock of the compiled driver, best of 30
"$GRAFT_REPO_ROOT/tools/create_proof_cpp" "$work/bundle17.bin" "$work/proof.bin" 30 > "$out/${tag}_create_proof_cpp.json"
cat "$out/${tag}_create_proof_cpp.json"
rm -rf "$work/bundle17.bin" "$work/prof_serial" "$work/prof_overlap"
