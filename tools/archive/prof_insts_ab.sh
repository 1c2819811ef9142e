# SQ_INSTS_VALU / SQ_WAVES of the 2^20 MSM's kernels for the two builds of the field products (v0 = -DSG_F29_ROW_SCAN, v1 = column chains)
set -e
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (relative paths below are removed and written under the repo copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
cp circuits_halo2_amd/libsumma_gpu.so /tmp/lib_orig.so
for v in v0 v1; do
  cp circuits_halo2_amd/libsumma_gpu_$v.so circuits_halo2_amd/libsumma_gpu.so
  rm -rf gpurun_out/prof_insts_$v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d gpurun_out/prof_insts_$v -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras > /dev/null 2> gpurun_out/prof_insts_$v.err; echo "$v rc=$?"
done
cp /tmp/lib_orig.so circuits_halo2_amd/libsumma_gpu.so
