set -e
mkdir -p gpurun_out/r03y
python bench.py --steps 20 --warmup 5 --batch-proofs 256 --batch-repeats 1 > gpurun_out/r03y/bench.json 2> gpurun_out/r03y/bench.err; echo "bench rc=$?"
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (relative paths below are removed and written under the repo copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG:-r03z} -- python3 bench.py --steps 20 --warmup 5 --no-cpu --batch-proofs 96 --batch-repeats 1 > gpurun_out/r03y/bench_under_rocprof.json 2> gpurun_out/r03y/rocprof.err; echo "rocprof rc=$?"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_${TAG:-r03z}_fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras > /dev/null 2>> gpurun_out/r03y/rocprof.err; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_${TAG:-r03z}_write -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras > /dev/null 2>> gpurun_out/r03y/rocprof.err; echo "write rc=$?"
rocprofv3 --pmc VALUBusy VALUUtilization -d gpurun_out/prof_${TAG:-r03z}_valu -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras > /dev/null 2>> gpurun_out/r03y/rocprof.err; echo "valu rc=$?"
python tools/summarize_prof.py ${TAG:-r03z}; python tools/summarize_valu.py ${TAG:-r03z} || true
# the condensed files travel back under gpurun_out/ (only that directory is merged); the raw databases stay on the box
cp profiles/${TAG:-r03z}_* gpurun_out/r03y/ 2>/dev/null || true
cp gpurun_out/r03y/bench.json gpurun_out/r03y/${TAG:-r03z}_bench_line.json
cp gpurun_out/r03y/bench_under_rocprof.json gpurun_out/r03y/${TAG:-r03z}_bench_line_under_rocprof.json
du -sh gpurun_out/prof_${TAG:-r03z}* | tail -5
rm -rf gpurun_out/prof_${TAG:-r03z} gpurun_out/prof_${TAG:-r03z}_fetch gpurun_out/prof_${TAG:-r03z}_write gpurun_out/prof_${TAG:-r03z}_valu
