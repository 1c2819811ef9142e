#!/bin/bash
# the driver's own shape (bench.py --steps 20 --warmup 5): the timed region's accumulations 11 .. 30 and what surrounds them
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (relative paths below are removed and written under the repo copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/msmshort
rocprofv3 --kernel-trace -d gpurun_out/msmshort -- python3 bench.py --no-extras --no-cpu --steps 20 --warmup 5 > gpurun_out/msmshort.json 2> gpurun_out/msmshort.err
python tools/msm_pipeline_trace.py gpurun_out/msmshort 11 31 > gpurun_out/msm_pipeline_short.txt 2>&1
python tools/msm_timeline_dump.py gpurun_out/msmshort 10 31 > gpurun_out/msm_timeline_short.txt 2>&1
rm -rf gpurun_out/msmshort
cat gpurun_out/msm_pipeline_short.txt | head -8
