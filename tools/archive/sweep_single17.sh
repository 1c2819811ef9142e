#!/bin/bash
# a single dense 2^17 fixed-base commitment (W, W' of a proof) and the 5-polynomial group (the quotient pieces): task length / lanes per task
set -euo pipefail
for p in "" "msm.log_seg=3" "msm.log_seg=4" "msm.log_seg=5" "msm.log_seg=3 msm.acc_threads=64" "msm.log_seg=4 msm.acc_threads=64" "msm.log_seg=4 msm.acc_threads=256"; do
  echo "== $p"; KS=17 MODES=fixed python tools/time_fixed_phases.py $p 2>&1 | grep -v amdgpu || exit 1
  M=5 python tools/run_fixed_batch.py $p 2>&1 | grep -v amdgpu || exit 1
done
