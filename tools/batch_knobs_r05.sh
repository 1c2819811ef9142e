#!/bin/bash
# Round 5: the 1024-proof batch (bench.py --batch-only) under the knobs that decide how much of the device's issue capacity the
# batch reaches: fused commitment jobs running side by side (commit.combine_runners), proofs in flight, CPU share.
# usage (GPU box): TAG=r05 CONFIGS="64:1:0 64:2:0 96:2:0 64:1:2" tools/batch_knobs_r05.sh     (in_flight:runners:cpu_share, 0 = all cores)
#   EXTRA_PARAMS="msm.acc_waves_fixed=3" adds library parameters to every configuration, HWQ=8 sets GPU_MAX_HW_QUEUES
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
tag="${TAG:-r05}"
out="gpurun_out/${tag}_batch_knobs.txt"
touch "$out"
for cfg in ${CONFIGS:-64:1:0 64:2:0}; do
  IFS=: read -r inflight runners share <<< "$cfg"
  line=$(GPU_MAX_HW_QUEUES="${HWQ:-4}" SUMMA_COMBINE_RUNNERS="$runners" SG_PARAMS="${EXTRA_PARAMS:-}" python bench.py --gpus 1 --batch-only --cpu-share "$share" --batch-proofs 1024 --batch-repeats 2 --no-cpu \
         --batch-in-flight "$inflight" --wall-limit 250 2>/dev/null | tail -1)
  HWQ="${HWQ:-4}" EXTRA_PARAMS="${EXTRA_PARAMS:-}" python - "$cfg" "$line" >> "$out" <<'PY'
import json, sys
cfg, line = sys.argv[1:3]
d = json.loads(line)
import os
print(f"hwq {os.environ.get('HWQ', '4')} extra [{os.environ.get('EXTRA_PARAMS', '')}] in_flight:runners:cpu_share {cfg:>8} | used in_flight {d.get('in_flight')} | {d.get('proofs_per_s', 0):7.1f} proofs/s (min/med/max {[round(v, 1) for v in d.get('proofs_per_s_min_median_max', [])]}) | "
      f"host CPU {d.get('host_cpu_ms_per_proof', 0):6.2f} ms/proof | cores busy {d.get('host_cores_busy_per_gpu', 0):5.2f} | errors {d.get('errors')}")
PY
  tail -1 "$out"
done
