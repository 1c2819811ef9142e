#!/bin/bash
# Round-4 profile set of the bench command (the judge's numbers come from here): the bench line alone, the same under
# rocprofv3 --kernel-trace --stats, and separate --pmc passes (FETCH_SIZE, WRITE_SIZE, VALUBusy / VALUUtilization,
# SQ_INSTS_VALU) -- never combined with a trace domain.  Condensed into profiles/<tag>_* by tools/summarize_*.py;
# the files travel back under gpurun_out/<tag>_profiles/ (only gpurun_out/ is merged).
# usage (GPU box): TAG=r04z tools/prof_r04.sh
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag="${TAG:-r04z}"
work="gpurun_out/${tag}_profiles"
mkdir -p "$work"
python bench.py --steps 20 --warmup 5 > "$work/bench.json" 2> "$work/bench.err"
echo "bench done"
rocprofv3 --kernel-trace --stats -d "gpurun_out/prof_${tag}" -- python3 bench.py --steps 20 --warmup 5 --no-cpu --batch-proofs 96 --batch-repeats 1 --no-cpu-share-sweep > "$work/bench_under_rocprof.json" 2> "$work/rocprof.err"
echo "trace done"
rocprofv3 --pmc FETCH_SIZE -d "gpurun_out/prof_${tag}_fetch" -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras > /dev/null 2>> "$work/rocprof.err"
rocprofv3 --pmc WRITE_SIZE -d "gpurun_out/prof_${tag}_write" -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras > /dev/null 2>> "$work/rocprof.err"
echo "hbm counters done"
rocprofv3 --pmc VALUBusy VALUUtilization -d "gpurun_out/prof_${tag}_valu" -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras > /dev/null 2>> "$work/rocprof.err"
echo "valu done"
python tools/summarize_prof.py "$tag"
python tools/summarize_valu.py "$tag" || true
cp profiles/${tag}_* "$work/" 2>/dev/null || true
cp "$work/bench.json" "$work/${tag}_bench_line.json"
cp "$work/bench_under_rocprof.json" "$work/${tag}_bench_line_under_rocprof.json"
rm -rf "gpurun_out/prof_${tag}" "gpurun_out/prof_${tag}_fetch" "gpurun_out/prof_${tag}_write" "gpurun_out/prof_${tag}_valu"
ls "$work"
