#!/bin/bash
# Round-5 profile set of the bench command (the judge's numbers come from here): the bench line alone, the same under
# rocprofv3 --kernel-trace --stats, and separate --pmc passes (FETCH_SIZE, WRITE_SIZE, VALUBusy / VALUUtilization,
# SQ_INSTS_VALU) -- never combined with a trace domain.  Condensed into profiles/<tag>_* by tools/summarize_*.py;
# the files travel back under gpurun_out/<tag>_profiles/ (only gpurun_out/ is merged).
# Round 5: every pass also writes the library's msm_accumulate launch log (bench.py --acc-log <dir>_acclog.json): tools/summarize_prof.py
# attributes each accumulation launch of the trace / counter pass to its job EXACTLY (record i = i-th chained launch), so the
# pipelined two-wave shape of the headline has a per-kernel figure of its own (`msm_accumulate_by_job`, `jobs_in_flight_at_issue`).
# usage (GPU box): TAG=r05z tools/prof_r05.sh
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag="${TAG:-r05z}"
work="gpurun_out/${tag}_profiles"
mkdir -p "$work"
t0=$(date +%s)
python bench.py --steps 20 --warmup 5 > "$work/bench.json" 2> "$work/bench.err"
echo "bench done in $(( $(date +%s) - t0 )) s (the driver's whole-run clock: headline + every extra)" | tee "$work/${tag}_bench_wall_clock.txt"
rocprofv3 --kernel-trace --stats -d "gpurun_out/prof_${tag}" -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --acc-log "gpurun_out/prof_${tag}_acclog.json" > "$work/bench_under_rocprof.json" 2> "$work/rocprof.err"
echo "trace done"
rocprofv3 --pmc FETCH_SIZE -d "gpurun_out/prof_${tag}_fetch" -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras --acc-log "gpurun_out/prof_${tag}_fetch_acclog.json" > /dev/null 2>> "$work/rocprof.err"
rocprofv3 --pmc WRITE_SIZE -d "gpurun_out/prof_${tag}_write" -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras --acc-log "gpurun_out/prof_${tag}_write_acclog.json" > /dev/null 2>> "$work/rocprof.err"
echo "hbm counters done"
rocprofv3 --pmc VALUBusy VALUUtilization -d "gpurun_out/prof_${tag}_valu" -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras --acc-log "gpurun_out/prof_${tag}_valu_acclog.json" > /dev/null 2>> "$work/rocprof.err"
echo "valu done"
python tools/summarize_prof.py "$tag"
python tools/summarize_valu.py "$tag" || true
cp profiles/${tag}_* "$work/" 2>/dev/null || true
cp "$work/bench.json" "$work/${tag}_bench_line.json"
cp "$work/bench_under_rocprof.json" "$work/${tag}_bench_line_under_rocprof.json"
rm -rf "gpurun_out/prof_${tag}" "gpurun_out/prof_${tag}_fetch" "gpurun_out/prof_${tag}_write" "gpurun_out/prof_${tag}_valu" gpurun_out/prof_${tag}*_acclog.json
ls "$work"
