#!/bin/bash
# whole proofs (compiled driver, k = 17, best of 30) under library tunables, with the per-phase times of the synchronised lap run
# usage: ab_proof_phases.sh "SG_PARAMS=a=1" ...
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_work
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
prover.export_bundle("gpurun_out/ab_work/bundle17.bin", params, pk, adv, c.instances()[0])
PY
run() { echo -n "$1 | "; env $1 ./tools/create_proof_cpp gpurun_out/ab_work/bundle17.bin gpurun_out/ab_work/proof.bin 30 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['create_proof_ms'], {k: v for k, v in d.items() if k[0].isdigit()})"; }
for knob in "$@"; do
  run "X=0"
  run "$knob"
done
rm -f gpurun_out/ab_work/bundle17.bin
