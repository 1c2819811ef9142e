# A/B of two library builds on one box, whole k = 17 proofs from the compiled prover (best of 30 per run, alternating):
# v1 = circuits_halo2_amd/libsumma_gpu_v1.so, new = the current build
set -e
out=gpurun_out/ab_proof; mkdir -p $out
cp circuits_halo2_amd/libsumma_gpu.so /tmp/lib_new.so
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
prover.export_bundle("gpurun_out/ab_proof/bundle17.bin", params, pk, adv, c.instances()[0])
PY
for r in 1 2 3; do for v in v1 new; do
  if [ $v = v1 ]; then cp circuits_halo2_amd/libsumma_gpu_v1.so circuits_halo2_amd/libsumma_gpu.so; else cp /tmp/lib_new.so circuits_halo2_amd/libsumma_gpu.so; fi
  echo -n "$v round $r | proof "
  ./tools/create_proof_cpp $out/bundle17.bin $out/proof_$v.bin 30 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['create_proof_ms'], {k:v for k,v in d.items() if k[0] in '123456'})"
done; done
cp /tmp/lib_new.so circuits_halo2_amd/libsumma_gpu.so
rm -f $out/bundle17.bin
