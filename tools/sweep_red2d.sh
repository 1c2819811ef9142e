#!/bin/bash
set -euo pipefail
for ms in 4 8 16 32; do for m in 5 8 16; do echo -n "max_sets=$ms "; M=$m python tools/run_fixed_batch.py msm.red2d_max_sets=$ms 2>&1 | grep -v amdgpu || exit 1; done; done
