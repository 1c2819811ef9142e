# kernel timeline of the sequential 2^20 MSM (tools/msm_timeline.py): run on the GPU box from the repo root
set -e
tag=${1:-r02t}
mkdir -p gpurun_out/$tag
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (relative paths below are removed and written under the repo copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -- python3 bench.py --steps 8 --warmup 3 --in-flight 1 --no-cpu --no-extras > gpurun_out/$tag/bench_seq.json 2> gpurun_out/$tag/rocprof.err
python tools/msm_timeline.py gpurun_out/prof_$tag 1048576 > gpurun_out/$tag/msm_timeline.txt
cat gpurun_out/$tag/msm_timeline.txt
rm -rf gpurun_out/prof_$tag
