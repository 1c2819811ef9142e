set -e
mkdir -p gpurun_out/r02f
./tools/microbench3 > gpurun_out/r02f/fp64_vs_f29.txt 2>&1 || true
cat gpurun_out/r02f/fp64_vs_f29.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err; echo "bench rc=$?"
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (relative paths below are removed and written under the repo copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02a -- python3 bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r02f/bench_under_rocprof.json 2> gpurun_out/r02f/rocprof.err; echo "rocprof rc=$?"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_r02a_fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras > /dev/null 2>> gpurun_out/r02f/rocprof.err; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_r02a_write -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras > /dev/null 2>> gpurun_out/r02f/rocprof.err; echo "write rc=$?"
rocprofv3 --pmc VALUBusy VALUUtilization -d gpurun_out/prof_r02a_valu -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras > /dev/null 2>> gpurun_out/r02f/rocprof.err; echo "valu rc=$?"
python tools/summarize_prof.py r02a; python tools/summarize_valu.py r02a || true
ls profiles | grep r02
du -sh gpurun_out/prof_r02a* | tail -5
rm -rf gpurun_out/prof_r02a/*/*_kernel_trace.csv.bak
