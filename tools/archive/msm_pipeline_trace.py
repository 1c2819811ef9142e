#!/usr/bin/env python3
"""Where the time of the headline's pipelined MSM steps goes: from a rocprofv3 --kernel-trace database of
`bench.py --no-extras --no-cpu --steps K`, the msm_accumulate launches of the steps issued from three host threads
(launch index FIRST..LAST), their durations, the gaps between one's end and the next one's start, what runs beside
them, and what runs in the gaps.  usage: msm_pipeline_trace.py <rocprof output dir> [first last]"""
import collections, glob, os, sqlite3, sys


def rows(directory):
    path = sorted(glob.glob(os.path.join(directory, "**", "*results.db"), recursive=True), key=os.path.getmtime)[-1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    cur.execute("select * from kernels")
    names = [d[0] for d in cur.description]
    return [dict(zip(names, r)) for r in cur.fetchall()]


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    rs = rows(sys.argv[1])
    rs.sort(key=lambda r: r["start"])
    acc = [r for r in rs if short(r["name"]).endswith("msm_accumulate")]
    first, last = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (20, 70)
    sel = acc[first:last]
    w0, w1 = sel[0]["start"], sel[-1]["end"]
    n = len(sel)
    print(f"{n} accumulate launches, window {(w1 - w0) / 1e6:.3f} ms = {(w1 - w0) / 1e3 / (n - 1) if n > 1 else 0:.1f} us per step")
    durs = [r["end"] - r["start"] for r in sel]
    print(f"accumulate duration: min {min(durs) / 1e3:.1f}  median {sorted(durs)[n // 2] / 1e3:.1f}  max {max(durs) / 1e3:.1f} us")
    gaps = [sel[i + 1]["start"] - sel[i]["end"] for i in range(n - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"gaps between consecutive accumulates: {len(pos)} positive, sum {sum(pos) / 1e3:.1f} us "
          f"({100.0 * sum(pos) / (w1 - w0):.1f} % of the window), overlaps {sum(1 for g in gaps if g <= 0)}")
    # time with 0 / 1 / 2 accumulates running
    ev = []
    for r in sel:
        ev += [(r["start"], 1), (r["end"], -1)]
    ev.sort()
    lvl, t_last, by = 0, w0, collections.Counter()
    for t, d in ev:
        by[lvl] += t - t_last
        t_last, lvl = t, lvl + d
    for k in sorted(by):
        print(f"  {k} accumulates running: {100.0 * by[k] / (w1 - w0):5.1f} %")
    # the other kernels: total duration inside the window, split by whether an accumulate was running at their start
    beside, alone = collections.Counter(), collections.Counter()
    spans = [(r["start"], r["end"]) for r in sel]
    import bisect
    starts = [s for s, _ in spans]
    for r in rs:
        if r["start"] < w0 or r["start"] >= w1 or short(r["name"]).endswith("msm_accumulate"):
            continue
        i = bisect.bisect_right(starts, r["start"]) - 1
        inside = i >= 0 and r["start"] < spans[i][1]
        (beside if inside else alone)[short(r["name"])] += r["end"] - r["start"]
    print("other kernels (us per step)        beside an accumulate   in a gap")
    for name in sorted(set(beside) | set(alone), key=lambda k: -(beside[k] + alone[k]))[:16]:
        print(f"  {name:38s} {beside[name] / 1e3 / n:9.1f} {alone[name] / 1e3 / n:12.1f}")
    print(f"  {'total':38s} {sum(beside.values()) / 1e3 / n:9.1f} {sum(alone.values()) / 1e3 / n:12.1f}")


if __name__ == "__main__":
    main()
