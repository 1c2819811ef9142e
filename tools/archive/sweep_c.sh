#!/bin/bash
set -euo pipefail
for lg in 14 15 16 17 18 19; do
for c in 10 12 13 14 15 16; do
  echo -n "2^$lg c=$c: "; SG_PARAMS=msm.window_bits=$c python bench.py --log-n $lg --steps 5 --warmup 1 --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['msm_phases_ms']; print(round(d['ms_per_step'],3), 'acc',round(p['accumulate_ms'],3),'red',round(p['reduce_ms'],3),'sort',round(p['sort_ms'],3), 'batched', round(d['batched']['ms_per_msm'],3))"
done; done
