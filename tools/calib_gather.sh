#!/bin/bash
# FETCH_SIZE against a known byte count for 64-byte gathers (tools/calib_gather.hip) -> gpurun_out/<tag>_fetch_calibration.json
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag="${1:-r04}"
[ -x tools/calib_gather ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/calib_gather.hip -o tools/calib_gather
out="gpurun_out/${tag}_fetch_calibration.json"
echo "[" > "$out"
first=1
for lp in 20 24; do
  rm -rf "gpurun_out/calib_$lp"
  line=$(rocprofv3 --pmc FETCH_SIZE -d "gpurun_out/calib_$lp" -- ./tools/calib_gather $lp 64 2>/dev/null | tail -1)
  python - "$lp" "$line" "$first" >> "$out" <<'PY'
import glob, json, sqlite3, sys
lp, line, first = sys.argv[1], json.loads(sys.argv[2]), sys.argv[3] == "1"
vals = []
for path in glob.glob(f"gpurun_out/calib_{lp}/*/*_results.db"):
    cur = sqlite3.connect(path).cursor()
    cur.execute("select kernel_name, counter_name, value from counters_collection")
    vals += [float(v) for k, c, v in cur.fetchall() if "gather64" in k and c == "FETCH_SIZE"]
line["FETCH_SIZE_KB_per_launch"] = sum(vals) / max(1, len(vals))
line["fetch_over_requested"] = line["FETCH_SIZE_KB_per_launch"] * 1024 / line["bytes_per_launch"]
print(("" if first else ",") + json.dumps(line))
PY
  first=0
  rm -rf "gpurun_out/calib_$lp"
done
echo "]" >> "$out"
cat "$out"
