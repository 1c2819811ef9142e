#!/usr/bin/env python3
"""Timeline of ONE proof of tools/create_proof_cpp from a rocprofv3 --kernel-trace database (not serialised: the real
overlap of the main stream, the side streams and the commitment jobs): start us, duration us, queue, stream, kernel, grid.
usage: proof_timeline_dump.py <rocprof output dir> [which proof from the end, default 2]"""
import glob, os, sqlite3, sys
path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*results.db"), recursive=True), key=os.path.getmtime)[-1]
cur = sqlite3.connect(path).cursor()
cur.execute("select * from kernels")
names = [d[0] for d in cur.description]
rows = sorted((dict(zip(names, r)) for r in cur.fetchall()), key=lambda r: r["start"])
short = lambda n: n.split("(")[0].replace("void ", "").replace("sg::", "")
starts = [i for i, r in enumerate(rows) if short(r["name"]) == "count_noncanonical_kernel"]      # first kernel of a proof (sanity checks on)
if len(starts) < 3:
    starts = [i for i, r in enumerate(rows) if short(r["name"]) == "fr_random_kernel" and int(r["grid_y"]) == 3]   # traces of round 3
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
seg = rows[starts[-which - 1]:starts[-which]]
t0 = seg[0]["start"]
print(f"# proof {len(starts) - which - 1} of {len(starts)}: {len(seg)} launches, span {(max(r['end'] for r in seg) - t0) / 1e3:.1f} us")
for r in seg:
    print(f"{(r['start'] - t0) / 1e3:8.1f} {(r['end'] - r['start']) / 1e3:7.1f}  q={r['queue_id']:<2} s={r['stream_id']:<2}  {short(r['name'])[:40]:40s} grid {r['grid_x']}x{r['grid_y']}")
