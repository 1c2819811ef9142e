#!/bin/bash
set -euo pipefail
for a in "msm.log_red_chunk=3" "msm.log_red_chunk=2" "msm.log_red_chunk=1" "msm.log_red_chunk=2 msm.red_threads=128" "msm.log_red_chunk=3 msm.red_threads=128" "msm.log_red_chunk=4"; do
  echo "== $a"; KS=20,17 MODES=generic python tools/time_fixed_phases.py $a 2>&1 | grep -v amdgpu || exit 1
done
