#!/bin/bash
set -euo pipefail
for m in 2 3 5 8 16; do for ls in 4 5 6 7; do echo -n "log_seg=$ls "; M=$m python tools/run_fixed_batch.py msm.log_seg=$ls 2>&1 | grep -v amdgpu || exit 1; done; done
