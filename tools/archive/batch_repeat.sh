#!/bin/bash
# the batch extra three times in a row (default hardware queues): stability of proofs/s at 1..4 proofs in flight
set -euo pipefail
for i in 1 2 3; do
  python bench.py --no-cpu --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['batch_k17']['by_in_flight']; print({k:round(v['proofs_per_s'],1) for k,v in b.items()}, 'msm', round(d['value']/1e6,1), round(d['sequential']['value']/1e6,1))" || exit 1
done
