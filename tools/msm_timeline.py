#!/usr/bin/env python3
"""Kernel timeline of one MSM from a rocprofv3 --kernel-trace directory.

    rocprofv3 --kernel-trace -d gpurun_out/prof_tl -- python3 bench.py --steps 8 --warmup 3 --in-flight 1 --no-cpu --no-extras
    python tools/msm_timeline.py gpurun_out/prof_tl 1048576 > profiles/rNN_msm_timeline.txt

Every MSM starts with an `msm_digits` launch over n threads; the launches up to the next one are that MSM's.  Printed:
per position in the sequence the kernel, its grid, the start offset from the MSM's first kernel and the duration,
averaged over the MSMs that have the same sequence (the modal one), and the gaps (device idle between kernels)."""
import collections
import glob
import os
import sqlite3
import sys


def rocpd_rows(directory, view):
    """rows of a view of rocprofv3's SQLite output (ROCm 7: `<dir>/<host>/<pid>_results.db`), as dicts"""
    out = []
    # gpurun merges every run's database into the same directory: only the newest one is the run being summarised
    for path in sorted(glob.glob(os.path.join(directory, "*", "*_results.db")), key=os.path.getmtime)[-1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        cur.execute(f"select * from {view}")
        names = [d[0] for d in cur.description]
        out += [dict(zip(names, r)) for r in cur.fetchall()]
    return out


def main():
    src, n = sys.argv[1], int(sys.argv[2])
    rows = sorted(rocpd_rows(src, "kernels"), key=lambda r: r["start"])
    jobs, cur = [], None
    for r in rows:
        name = r["name"].split("(")[0]
        grid = int(r["grid_x"]) * int(r["grid_y"]) * int(r["grid_z"])
        if name == "sg::msm_digits":
            if cur:
                jobs.append(cur)
            cur = [] if grid == n else None
        if cur is not None:
            cur.append((name, grid, r["start"], r["end"]))
    if cur:
        jobs.append(cur)
    shapes = collections.Counter(tuple((k[0], k[1]) for k in j) for j in jobs)
    if not shapes:
        print("no MSM of that size in the trace")
        return
    shape, count = shapes.most_common(1)[0]
    same = [j for j in jobs if tuple((k[0], k[1]) for k in j) == shape]
    print(f"{len(jobs)} MSMs of {n} points in the trace, {count} with the modal sequence of {len(shape)} launches")
    print(f"{'#':>3} {'kernel':40} {'grid':>9} {'start us':>10} {'dur us':>9} {'gap before us':>14}")
    busy = 0.0
    for i, (name, grid) in enumerate(shape):
        st = sum(j[i][2] - j[0][2] for j in same) / len(same) / 1e3
        du = sum(j[i][3] - j[i][2] for j in same) / len(same) / 1e3
        gap = sum((j[i][2] - j[i - 1][3]) for j in same) / len(same) / 1e3 if i else 0.0
        busy += du
        print(f"{i:3d} {name[:40]:40} {grid:9d} {st:10.1f} {du:9.1f} {gap:14.1f}")
    span = sum(j[-1][3] - j[0][2] for j in same) / len(same) / 1e3
    print(f"span {span:.1f} us, kernels busy {busy:.1f} us, idle inside {span - busy:.1f} us")
    by = collections.defaultdict(float)
    for i, (name, grid) in enumerate(shape):
        by[name] += sum(j[i][3] - j[i][2] for j in same) / len(same) / 1e3
    for name, us in sorted(by.items(), key=lambda kv: -kv[1]):
        print(f"  {name:40} {us:9.1f} us  {100 * us / busy:5.1f} %")


if __name__ == "__main__":
    main()
