#!/bin/bash
# proofs per second of the batch extra under different numbers of HIP hardware queues (GPU_MAX_HW_QUEUES)
set -euo pipefail
for q in 4 8 16 32; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['batch_k17']['by_in_flight']; print({k:round(v['proofs_per_s'],1) for k,v in b.items()}, 'msm', round(d['value']/1e6,1))" || exit 1
done
