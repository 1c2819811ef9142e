# A/B of two library builds on one box for the small kernels: v1 = circuits_halo2_amd/libsumma_gpu_v1.so, new = the current build
set -e
cp circuits_halo2_amd/libsumma_gpu.so /tmp/lib_new.so
for r in 1 2; do for v in v1 new; do
  if [ $v = v1 ]; then cp circuits_halo2_amd/libsumma_gpu_v1.so circuits_halo2_amd/libsumma_gpu.so; else cp /tmp/lib_new.so circuits_halo2_amd/libsumma_gpu.so; fi
  echo "== $v round $r"
  python tools/time_batch_invert.py 2>/dev/null | tr '\n' ' '; echo
  python bench.py --steps 30 --warmup 5 --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('msm', round(d['ms_per_step'],3), 'sequential', round(d['sequential']['ms_per_step'],3), {k:round(v,3) for k,v in d['msm_phases_ms'].items() if k.endswith('_ms')})"
done; done
cp /tmp/lib_new.so circuits_halo2_amd/libsumma_gpu.so
