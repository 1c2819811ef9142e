#!/bin/bash
# Round 5: kernel timeline of ONE blocking 2^20 MSM (what an unmodified halo2 sees per best_multiexp call): the launches of an MSM in
# order, their start offsets and durations, the gaps between them (tools/msm_timeline.py over a kernel trace of bench.py --in-flight 1)
# usage (GPU box): TAG=r05 tools/prof_msm_timeline_r05.sh      -> gpurun_out/<tag>_msm_timeline_sequential.txt
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag="${TAG:-r05}"
rm -rf "gpurun_out/prof_tl_$tag"
rocprofv3 --kernel-trace -d "gpurun_out/prof_tl_$tag" -- python3 bench.py --steps 12 --warmup 3 --in-flight 1 --no-cpu --no-extras > "gpurun_out/${tag}_msm_timeline_line.json" 2> "gpurun_out/${tag}_msm_timeline.err"
python tools/msm_timeline.py "gpurun_out/prof_tl_$tag" 1048576 > "gpurun_out/${tag}_msm_timeline_sequential.txt"
rm -rf "gpurun_out/prof_tl_$tag"
tail -40 "gpurun_out/${tag}_msm_timeline_sequential.txt"
