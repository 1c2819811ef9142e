#!/bin/bash
set -euo pipefail
for q in 0 2; do for m in 3 4 5 6 8 12; do echo -n "quad=$q "; M=$m python tools/run_fixed_batch.py msm.quad=$q 2>&1 | grep -v amdgpu || exit 1; done; done
