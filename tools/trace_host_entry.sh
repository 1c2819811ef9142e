#!/bin/bash
# usage: trace_host_entry.sh <tag> <commit|msm> [SG_PARAMS]: kernel + memory-copy trace of the host-pointer entry point, last call's timeline
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag="$1"; which="$2"; export SG_PARAMS="${3:-}"
rm -rf "gpurun_out/hosttrace_$tag"
rocprofv3 --kernel-trace --memory-copy-trace -d "gpurun_out/hosttrace_$tag" -- python3 tools/trace_host_entry.py "$which" > "gpurun_out/hosttrace_$tag.log" 2> "gpurun_out/hosttrace_$tag.err"
cat "gpurun_out/hosttrace_$tag.log"
python - "gpurun_out/hosttrace_$tag" > "gpurun_out/host_timeline_$tag.txt" <<'PY'
import glob, os, sqlite3, sys
path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*results.db"), recursive=True), key=os.path.getmtime)[-1]
db = sqlite3.connect(path); cur = db.cursor()
def rows(view):
    cur.execute(f"select * from {view}")
    names = [d[0] for d in cur.description]
    return [dict(zip(names, r)) for r in cur.fetchall()]
ev = [(r["start"], r["end"], "K", r["name"].split("(")[0].replace("void ", "").replace("sg::", ""), f"q={r['queue_id']} grid {r['grid_x']}x{r['grid_y']}") for r in rows("kernels")]
try:
    for r in rows("memory_copies"):
        ev.append((r["start"], r["end"], "C", str(r.get("name", "copy")), f"{r.get('size', 0) / 2**20:.1f} MiB"))
except Exception as ex:
    print("# no memory copy view:", ex)
ev.sort()
# the last MSM call = from the last-but-K msm_digits on: take the final 1/6 of the kernels after the last gap > 2 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 40):i]) > 1.0e6:
        cut = i
seg = ev[cut:]
t0 = seg[0][0]
print(f"# {len(seg)} events, span {(max(e[1] for e in seg) - t0) / 1e3:.1f} us")
for s_, e_, kind, name, extra in seg:
    print(f"{(s_ - t0) / 1e3:9.1f} {(e_ - s_) / 1e3:8.1f}  {kind}  {name[:40]:40s} {extra}")
PY
rm -rf "gpurun_out/hosttrace_$tag"
head -80 "gpurun_out/host_timeline_$tag.txt"
