#!/bin/bash
set -euo pipefail
for p in 22 23 24 25 26 27; do
  echo "== fuse 2^$p"; SG_PARAMS=msm.log_fuse_entries=$p python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['batched'], d['proof_oplist_k17']['msm16x2^17_ms'])"
done
