#!/bin/bash
# persistent msm_accumulate: waves per SIMD (msm.acc_waves: 8 = one ticket per wave / grid = tasks, 3 = full file, 2 = room for others, 0 = auto)
# against steps in flight; headline points/s, one-at-a-time ms, accumulate ms
set -euo pipefail
for rep in 1 2; do
for w in 8 3 2 0; do
  for f in 3 4; do
    SG_PARAMS=msm.acc_waves=$w python bench.py --no-extras --no-cpu --steps 60 --in-flight $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('acc_waves=$w in_flight=$f  %.1f M points/s  step %.3f ms  sequential %.3f ms  accumulate %.3f ms  reduce %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['sequential']['ms_per_step'], d['msm_phases_ms']['accumulate_ms'], d['msm_phases_ms']['reduce_ms']))"
  done
done
done
