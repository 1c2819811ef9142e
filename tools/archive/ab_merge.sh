# whole-proof A/B of the merge rounds' lane mapping (msm.merge_quad_tasks: rounds with more tasks use one lane per addition), k = 17, best of 30 per run
set -e
out=gpurun_out/ab_merge; mkdir -p $out
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
prover.export_bundle("gpurun_out/ab_merge/bundle17.bin", params, pk, adv, c.instances()[0])
PY
run() { echo -n "$1 | "; env $1 ./tools/create_proof_cpp $out/bundle17.bin $out/proof.bin 30 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['create_proof_ms'], {k:v for k,v in d.items() if k[0] in '1346'})"; }
for r in 1 2 3; do
  run "X=0"
  run "SG_PARAMS=msm.merge_quad_tasks=0"
  run "SG_PARAMS=msm.merge_quad_tasks=100000"
done
rm -f $out/bundle17.bin
