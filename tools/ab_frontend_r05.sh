#!/bin/bash
# A/B of the MSM front end on ONE box: blocking 2^20 MSMs (bench.py --in-flight 1 --no-extras) with the scans inside the sort's kernels
# (msm.fused_frontend=1, round 5) and as launches of their own (0): step time and the phases from the MSM's own HIP events
# usage (GPU box): tools/ab_frontend_r05.sh <tag>
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
tag="${1:-r05}"
out="gpurun_out/${tag}_ab_frontend.txt"
: > "$out"
for rep in 1 2; do
  for fe in 0 1; do
    line=$(SG_PARAMS="msm.fused_frontend=$fe" python bench.py --steps 20 --warmup 5 --no-cpu --no-extras 2>/dev/null | tail -1)
    python - "$fe" "$line" >> "$out" <<'PY'
import json, sys
fe, line = sys.argv[1:3]
d = json.loads(line)
p = d["msm_phases_ms"]
print(f"fused_frontend {fe} | {d['value'] / 1e6:6.1f} M points/s ({d['ms_per_step']:.3f} ms/step, three in flight) | one at a time {d['sequential']['ms_per_step']:.3f} ms | "
      f"digits {p['digits_ms']:.3f} sort {p['sort_ms']:.3f} order {p['order_ms']:.3f} accumulate {p['accumulate_ms']:.3f} reduce {p['reduce_ms']:.3f} total {p['total_ms']:.3f} | "
      f"pipelined accumulate launch {d['roofline'].get('launch_ms', 0):.3f} ms")
PY
    tail -1 "$out"
  done
done
