#!/bin/bash
# kernel trace of the 1024-proof batch (bench.py --batch-only, IN_FLIGHT (default 64) in flight): concurrency and kernel time by kernel in the steady state
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag="${1:-r04}"
rm -rf "gpurun_out/prof_batch_$tag"
rocprofv3 --kernel-trace -d "gpurun_out/prof_batch_$tag" -- python3 bench.py --gpus 1 --batch-only --batch-proofs 1024 --batch-repeats 1 --no-cpu --batch-in-flight ${IN_FLIGHT:-64} --wall-limit 300 > "gpurun_out/${tag}_batch_under_rocprof.json" 2> "gpurun_out/${tag}_batch_rocprof.err"
python tools/batch_concurrency.py "gpurun_out/prof_batch_$tag" 0.6 0.95 > "gpurun_out/${tag}_batch_concurrency.txt"
rm -rf "gpurun_out/prof_batch_$tag"
cat "gpurun_out/${tag}_batch_concurrency.txt"
