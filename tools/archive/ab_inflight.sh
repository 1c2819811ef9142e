#!/bin/bash
# usage: ab_inflight.sh <steps> "<in-flight list>" cfg1 cfg2 ...
set -euo pipefail
steps=$1; fl=$2; shift; shift
for rep in 1 2; do
for cfg in "$@"; do
 for f in $fl; do
  SG_PARAMS=$cfg python bench.py --no-extras --no-cpu --steps $steps --in-flight $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg in_flight=$f  %.1f M points/s  step %.3f ms  sequential %.3f ms  accumulate %.3f ms  reduce %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['sequential']['ms_per_step'], d['msm_phases_ms']['accumulate_ms'], d['msm_phases_ms']['reduce_ms']))"
 done
done
done
