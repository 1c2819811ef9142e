#!/bin/bash
set -euo pipefail
for lg in 20 17; do
for p in "" "msm.red_threads=128" "msm.red_threads=128,msm.log_red_chunk=2" "msm.red_threads=64,msm.log_red_chunk=3" "msm.red_threads=64,msm.log_red_chunk=2" "msm.red_threads=128,msm.log_red_chunk=4" "msm.red_threads=256,msm.log_red_chunk=2"; do
  echo "== 2^$lg $p"; SG_PARAMS=$p python bench.py --log-n $lg --steps 5 --warmup 1 --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['msm_phases_ms'].items() if k in ('accumulate_ms','reduce_ms','total_ms')})"
done; done
