#!/bin/bash
set -euo pipefail
for m in 1 2 3 5 8; do M=$m python tools/run_fixed_batch.py "$@" || exit 1; done
