#!/usr/bin/env python3
"""Which hardware queue each HIP stream of a traced run landed on, and what ran there: from a rocprofv3 --kernel-trace
database.  usage: stream_queue_map.py <rocprof output dir>"""
import collections, glob, os, sqlite3, sys
path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*results.db"), recursive=True), key=os.path.getmtime)[-1]
cur = sqlite3.connect(path).cursor()
cur.execute("select * from kernels")
names = [d[0] for d in cur.description]
rs = [dict(zip(names, r)) for r in cur.fetchall()]
short = lambda n: n.split("(")[0].replace("void ", "").replace("sg::", "")
by = collections.defaultdict(collections.Counter)
tot = collections.Counter()
for r in rs:
    key = (r["queue_id"], r["stream_id"])
    by[key][short(r["name"])] += 1
    tot[key] += r["end"] - r["start"]
for key in sorted(by):
    top = ", ".join(f"{n} x{c}" for n, c in by[key].most_common(4))
    print(f"queue {key[0]}  stream {key[1]:3d}  {tot[key] / 1e6:8.2f} ms of kernels   {top}")
