#!/bin/bash
set -euo pipefail
for a in "msm.quad=1" "msm.quad=2" "msm.quad=0 msm.log_red_chunk=2" "msm.quad=0 msm.log_red_chunk=4" "msm.quad=2 msm.log_red_chunk=4" "msm.quad=0 msm.red_threads=128" "msm.quad=0 msm.red_threads=64" "msm.log_seg=7" "msm.log_seg=6"; do
  echo "== $a"; python tools/run_fixed_batch.py $a || exit 1
done
