#!/bin/bash
# the gate interpreter on the reference circuit's program: rows per workgroup / reload distance (development knobs of csrc/gates.hip)
set -euo pipefail
for env in "" "SG_GATES_ROWS=64" "SG_GATES_ROWS=128" "SG_GATES_ROWS=256" "SG_GATES_RELOAD=8" "SG_GATES_RELOAD=16" "SG_GATES_RELOAD=48" "SG_GATES_RELOAD=48 SG_GATES_CONVERT=1"; do
  echo "== $env"; env $env SG_GATES_DEBUG=1 KS=17 python tools/time_mst_gates.py 2>&1 | grep -E "gates:|k=17" | sort -u | head -3 || exit 1
done
