#!/bin/bash
# usage: trace_msm_pipeline.sh <tag> <SG_PARAMS> [in_flight]: kernel trace of the headline's pipelined steps -> gpurun_out/msm_pipeline_<tag>.txt
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (relative paths below are removed and written under the repo copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/msmtrace_$1
SG_PARAMS=$2 rocprofv3 --kernel-trace -d gpurun_out/msmtrace_$1 -- python3 bench.py --no-extras --no-cpu --steps 60 --in-flight ${3:-3} > gpurun_out/msmtrace_$1.json 2> gpurun_out/msmtrace_$1.err
python tools/msm_pipeline_trace.py gpurun_out/msmtrace_$1 20 70 > gpurun_out/msm_pipeline_$1.txt 2>&1
python tools/msm_timeline_dump.py gpurun_out/msmtrace_$1 30 34 > gpurun_out/msm_timeline_$1.txt 2>&1
rm -rf gpurun_out/msmtrace_$1
echo "== $1 ($2)"; cat gpurun_out/msm_pipeline_$1.txt
