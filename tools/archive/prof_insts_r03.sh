# hardware instruction counts of the MSM kernels at the end of round 3 (persistent accumulation): SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVES
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (relative paths below are removed and written under the repo copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_insts_r03
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d gpurun_out/prof_insts_r03 -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras > /dev/null 2> gpurun_out/prof_insts_r03.err; echo "rc=$?"
python - <<'PY'
import collections, glob, json, sqlite3
rows = []
for path in glob.glob("gpurun_out/prof_insts_r03/*/*_results.db"):
    cur = sqlite3.connect(path).cursor()
    cur.execute("select kernel_name, grid_size, queue_id, counter_name, value, start from counters_collection")
    rows += cur.fetchall()
per = collections.defaultdict(lambda: collections.defaultdict(list))
last_digits = {}
for name, grid, queue, counter, value, start in sorted(rows, key=lambda r: r[5]):
    short = name.split('(')[0].replace('void ', '')
    if short == "sg::msm_digits":
        last_digits[queue] = grid            # the job a later kernel of this queue belongs to: 1048576 = one 2^20 MSM
    key = f"{short}@grid{grid}"
    if short != "sg::msm_digits":
        key += f"@job{last_digits.get(queue, 0)}"
    per[key][counter].append(float(value))
out = {k: dict({c: round(sum(v) / len(v)) for c, v in d.items()}, launches=len(next(iter(d.values())))) for k, d in sorted(per.items()) if "msm_" in k}
adds = 16 * (1 << 20)
note = {}
for k, v in out.items():
    if k.startswith("sg::msm_accumulate") and k.endswith("@job1048576") and "SQ_INSTS_VALU" in v:
        note[k] = round(v["SQ_INSTS_VALU"] / (adds / 64), 1)
json.dump({"note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES of bench.py --steps 10 --warmup 2 --no-cpu --no-extras (tools/prof_insts_r03.sh), "
                   "averages per launch; msm_accumulate is the persistent launch (grid 196608 = three waves per SIMD, 131072 = two): 16 * 2^20 bucket additions each",
           "valu_wave_instructions_per_64_additions": note, "kernels": out}, open("gpurun_out/r03z_insts.json", "w"), indent=1)
print(json.dumps(note))
PY
rm -rf gpurun_out/prof_insts_r03
