#!/bin/bash
# usage: trace_proof_queues.sh <tag> [ENV=VAL ...]: hardware queue of every stream of the compiled prover (3 proofs at k = 17)
tag=$1; shift
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
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (relative paths below are removed and written under the repo copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/pq_$tag
rocprofv3 --kernel-trace -d gpurun_out/pq_$tag -- ./tools/create_proof_cpp gpurun_out/r03g/bundle17.bin gpurun_out/r03g/proof.bin 6 > gpurun_out/pq_$tag.json 2> gpurun_out/pq_$tag.err
echo "== $tag $@"; cat gpurun_out/pq_$tag.json | head -c 300; echo
python tools/stream_queue_map.py gpurun_out/pq_$tag
python tools/proof_timeline_dump.py gpurun_out/pq_$tag > gpurun_out/proof_timeline_$tag.txt 2>&1
rm -rf gpurun_out/pq_$tag
