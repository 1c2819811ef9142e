#!/usr/bin/env python3
"""Per-proof kernel totals from a rocprofv3 --kernel-trace directory of tools/create_proof_cpp (tools/prof_proof.sh):
a proof = the kernels between the first blinding draws of two consecutive proofs."""
import collections, sys
from msm_timeline import rocpd_rows

src = sys.argv[1]
rows = sorted(rocpd_rows(src, "kernels"), key=lambda r: r["start"])
name = lambda r: r["name"].split("(")[0]
grid = lambda r: int(r["grid_x"]) * int(r["grid_y"]) * int(r["grid_z"])
starts = []
# round 4: the compiled driver runs with sanity checks, whose range check of the advice columns (count_noncanonical_kernel, grid.y = 3)
# is the FIRST kernel of every proof -- and the blinding rows of phase 3 became a second 256 x 3 launch of fr_random_kernel
marker = [i for i, r in enumerate(rows) if name(r) == "sg::count_noncanonical_kernel"]
if len(marker) >= 3:
    starts = marker
for i in range(len(rows) - 3 if not starts else 0):
    # round 3: the blinding rows of the three advice columns are ONE launch with grid.y = 3 (sg_fr_random_batch_dev)
    if name(rows[i]) == "sg::fr_random_kernel" and int(rows[i]["grid_y"]) == 3:
        starts.append(i)
    elif all(name(rows[i + d]) == "sg::fr_random_kernel" for d in range(4)) and grid(rows[i + 3]) > 4096 and (i == 0 or name(rows[i - 1]) != "sg::fr_random_kernel"):
        starts.append(i)   # (traces of earlier rounds: three single launches open phase 1)
print(f"# {len(starts)} proof starts found")
for pi in range(max(0, len(starts) - 4), len(starts) - 1):
    seg = rows[starts[pi]:starts[pi + 1]]
    span = (seg[-1]["end"] - seg[0]["start"]) / 1e6
    total = sum(r["end"] - r["start"] for r in seg) / 1e6
    iv = sorted((r["start"], r["end"]) for r in seg)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"proof {pi}: span {span:.2f} ms, {len(seg)} kernel launches, sum of kernel durations {total:.2f} ms, GPU busy {busy / 1e6:.2f} ms")
seg = rows[starts[-2]:starts[-1]]
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    a = agg[name(r)]
    a[0] += 1
    a[1] += r["end"] - r["start"]
print("# kernels of the last complete proof, by total duration")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k[:44]:44} {c:4d} launches {t / 1e3:9.1f} us")
