#!/bin/bash
# A/B two builds of the library in one GPU session (same device): bench each twice, interleaved
set -euo pipefail
cd circuits_halo2_amd
cp libsumma_gpu.so /tmp/lib_orig.so
for r in 1 2; do for v in v0 v1; do cp libsumma_gpu_$v.so libsumma_gpu.so; echo "== $v round $r"; (cd .. && python bench.py --steps 10 --warmup 2 --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['msm_phases_ms'].items()})"); done; done
cp /tmp/lib_orig.so libsumma_gpu.so
