#!/bin/bash
# the host's own timeline of one k = 17 proof of the compiled driver (SG_PROVER_TRACE=1: when each step of the driver was reached,
# no synchronisation added) -- where the wall clock of a lone proof goes between the kernels
# usage (GPU box): TAG=r05t tools/proof_host_trace_r05.sh
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
tag="${TAG:-r05t}"
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
PY
SG_PROVER_TRACE=1 tools/create_proof_cpp "$work/bundle17.bin" "$work/proof.bin" 12 > "$work/out.json" 2> "$work/trace.txt" || true
# the marks of the LAST proof of the run
python - "$work/trace.txt" <<'PY' | tee "gpurun_out/${tag}_proof_host_trace.txt"
import sys
lines = [l.rstrip("\n") for l in open(sys.argv[1]) if " us (+" in l]
starts = [i for i, l in enumerate(lines) if l.strip().startswith("0.0 us") or "(+    0.0)" in l and i == 0]
# a proof's marks start where the time goes back
cut = [0]
prev = -1.0
for i, l in enumerate(lines):
    t = float(l.split("us")[0])
    if t < prev:
        cut.append(i)
    prev = t
last = lines[cut[-1]:]
print("\n".join(last))
PY
cat "$work/out.json"
rm -rf "$work"
