# A/B of two library builds on one box with the whole bench (MSM, NTT, proofs, batch): v1 = circuits_halo2_amd/libsumma_gpu_v1.so,
# new = the current build; alternating, two rounds
set -e
cp circuits_halo2_amd/libsumma_gpu.so /tmp/lib_new.so
for r in 1 2; do for v in v1 new; do
  if [ $v = v1 ]; then cp circuits_halo2_amd/libsumma_gpu_v1.so circuits_halo2_amd/libsumma_gpu.so; else cp /tmp/lib_new.so circuits_halo2_amd/libsumma_gpu.so; fi
  echo -n "$v round $r | "
  python bench.py --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('msm', round(d['ms_per_step'],3), 'ntt22', round(d['ntt']['2^22']['ms'],4), 'ntt17', round(d['ntt']['2^17']['ms'],4), 'proof py', round(d['create_proof_k17']['ms'],3), 'cpp', d['create_proof_k17']['ms_cpp_driver'], 'k13 cpp', d['full_flow_levels20_k13']['create_proof_ms_cpp_driver'], 'k11 cpp', d['reference_circuit_k11']['create_proof_ms_cpp_driver'], 'batch', {k:round(v['proofs_per_s'],1) for k,v in d['batch_k17']['by_in_flight'].items()}, 'mst', round(d['witness_mst_2^20']['ms'],2))"
done; done
cp /tmp/lib_new.so circuits_halo2_amd/libsumma_gpu.so
