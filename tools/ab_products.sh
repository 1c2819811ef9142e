# A/B of the two spellings of the field products (bn254_f29.cuh) on one box: the library built with -DSG_F29_ROW_SCAN
# (circuits_halo2_amd/libsumma_gpu_v0.so: plain C++, as the compiler schedules it) against the default build
# (libsumma_gpu_v1.so: column chains), alternating; the headline MSM with its phases, one k = 17 proof from the compiled
# prover (best of 30), and at the end the batch figure of the bench for each
set -e
out=gpurun_out/ab_products; mkdir -p $out
cd circuits_halo2_amd; cp libsumma_gpu.so /tmp/lib_orig.so; cd ..
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
prover.export_bundle("gpurun_out/ab_products/bundle17.bin", params, pk, adv, c.instances()[0])
PY
for r in 1 2 3; do for v in v0 v1; do
  cp circuits_halo2_amd/libsumma_gpu_$v.so circuits_halo2_amd/libsumma_gpu.so
  echo -n "$v round $r | msm "
  python bench.py --steps 30 --warmup 5 --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'sequential', round(d['sequential']['ms_per_step'],3), {k:round(v,3) for k,v in d['msm_phases_ms'].items() if k.endswith('_ms')}, end=' | proof ')"
  ./tools/create_proof_cpp $out/bundle17.bin $out/proof_$v.bin 30 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['create_proof_ms'])"
done; done
for v in v0 v1; do
  cp circuits_halo2_amd/libsumma_gpu_$v.so circuits_halo2_amd/libsumma_gpu.so
  echo -n "$v full bench | "
  python bench.py --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('msm', round(d['ms_per_step'],3), 'ntt22', round(d['ntt']['2^22']['ms'],4), 'ntt17', round(d['ntt']['2^17']['ms'],4), 'proof py', round(d['create_proof_k17']['ms'],3), 'cpp', d['create_proof_k17']['ms_cpp_driver'], 'batch', {k:round(v['proofs_per_s'],1) for k,v in d['batch_k17']['by_in_flight'].items()}, 'mst', round(d['witness_mst_2^20']['ms'],2))"
done
cp /tmp/lib_orig.so circuits_halo2_amd/libsumma_gpu.so
rm -f $out/bundle17.bin
