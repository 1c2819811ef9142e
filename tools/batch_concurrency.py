#!/usr/bin/env python3
"""How much of the GPU's time a batch of proofs in flight actually overlaps: from a rocprofv3 --kernel-trace database,
the time-weighted distribution of the number of kernels running at once, the kernels that run ALONE for the longest
total time, and the use of the hardware queues.  usage: batch_concurrency.py <rocprof output dir> [t0_fraction t1_fraction]"""
import collections, glob, os, sqlite3, sys


def rows(directory):
    path = sorted(glob.glob(os.path.join(directory, "*", "*_results.db")), key=os.path.getmtime)[-1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    cur.execute("select * from kernels")
    names = [d[0] for d in cur.description]
    return [dict(zip(names, r)) for r in cur.fetchall()]


def main():
    rs = rows(sys.argv[1])
    rs.sort(key=lambda r: r["start"])
    t_lo, t_hi = rs[0]["start"], max(r["end"] for r in rs)
    f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.5, 0.95)
    w0, w1 = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)       # steady-state window of the run
    ev = []
    for i, r in enumerate(rs):
        s, e = max(r["start"], w0), min(r["end"], w1)
        if e > s:
            ev.append((s, 1, i))
            ev.append((e, -1, i))
    ev.sort()
    live, last = set(), w0
    by_level = collections.Counter()
    alone = collections.Counter()
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            by_level[len(live)] += dt
            if len(live) == 1:
                alone[rs[next(iter(live))]["name"].split("(")[0]] += dt
        last = t
        if d > 0:
            live.add(i)
        else:
            live.discard(i)
    by_level[0] += max(0, w1 - last)
    total = w1 - w0
    print(f"window {total / 1e6:.2f} ms, {sum(1 for r in rs if w0 <= r['start'] < w1)} kernel launches in it")
    for lvl in sorted(by_level):
        print(f"  {lvl} kernels running: {100.0 * by_level[lvl] / total:5.1f} % of the time")
    durs = sum(min(r["end"], w1) - max(r["start"], w0) for r in rs if min(r["end"], w1) > max(r["start"], w0))
    print(f"sum of kernel durations / wall = {durs / total:.2f} (average number of kernels running)")
    print("kernels that run ALONE, by total time alone:")
    for name, t in alone.most_common(14):
        print(f"  {name:44s} {t / 1e3:9.1f} us  {100.0 * t / total:5.1f} % of the window")
    tot = collections.Counter()
    cnt = collections.Counter()
    for r in rs:
        if w0 <= r["start"] < w1:
            name = r["name"].split("(")[0]
            tot[name] += r["end"] - r["start"]
            cnt[name] += 1
    print("kernels by total duration in the window (concurrent kernels both count):")
    for name, t in tot.most_common(24):
        print(f"  {name:44s} {t / 1e3:10.1f} us  {cnt[name]:6d} launches  {100.0 * t / total:5.1f} % of the window")
    q = collections.Counter()
    for r in rs:
        if w0 <= r["start"] < w1:
            q[r.get("queue_id", r.get("queue", 0))] += r["end"] - r["start"]
    print("kernel time by hardware queue:", {k: round(v / 1e6, 2) for k, v in sorted(q.items())})


if __name__ == "__main__":
    main()
