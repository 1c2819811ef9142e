#!/bin/bash
# sweep msm.log_scatter_rounds: fused fixed-base batch (k=17 x16) and single generic commits (k=17, 20)
set -euo pipefail
for r in 0 1 2 3 4 5; do
  echo "== log_scatter_rounds=$r"
  python tools/run_fixed_batch.py msm.log_scatter_rounds=$r || exit 1
  MODES=generic python tools/time_fixed_phases.py msm.log_scatter_rounds=$r || exit 1
done
