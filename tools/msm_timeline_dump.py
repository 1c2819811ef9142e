#!/usr/bin/env python3
"""Text timeline of a few pipelined MSM steps from a rocprofv3 --kernel-trace database: every launch between the
start of msm_accumulate launch FIRST and the end of launch LAST: start (us from the window's start), duration, queue, name.
usage: msm_timeline_dump.py <rocprof output dir> [first last]"""
import glob, os, sqlite3, sys


def main():
    path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*results.db"), recursive=True), key=os.path.getmtime)[-1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    cur.execute("select * from kernels")
    names = [d[0] for d in cur.description]
    rs = [dict(zip(names, r)) for r in cur.fetchall()]
    rs.sort(key=lambda r: r["start"])
    short = lambda n: n.split("(")[0].replace("void ", "").replace("sg::", "")
    acc = [r for r in rs if short(r["name"]).endswith("msm_accumulate")]
    first, last = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (30, 34)
    w0, w1 = acc[first]["start"], acc[last]["end"]
    qkey = "queue_id" if "queue_id" in names else ("queue" if "queue" in names else None)
    skey = "stream_id" if "stream_id" in names else None
    print("columns:", names)
    for r in rs:
        if r["end"] < w0 or r["start"] > w1:
            continue
        print(f"{(r['start'] - w0) / 1e3:9.1f} {(r['end'] - r['start']) / 1e3:8.1f}  q={r.get(qkey)} s={r.get(skey)} tid={r.get('tid')}  {short(r['name'])}")


if __name__ == "__main__":
    main()
