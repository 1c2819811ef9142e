#!/bin/bash
# usage: ab_quick_headline.sh <steps> cfg1 cfg2 ... : headline line fields for each SG_PARAMS setting, in-flight 3, twice
set -euo pipefail
steps=$1; shift
for rep in 1 2; do
for cfg in "$@"; do
  SG_PARAMS=$cfg python bench.py --no-extras --no-cpu --steps $steps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg  %.1f M points/s  step %.3f ms  sequential %.3f ms  accumulate %.3f ms  reduce %.3f ms  sort %.3f' % (d['value']/1e6, d['ms_per_step'], d['sequential']['ms_per_step'], d['msm_phases_ms']['accumulate_ms'], d['msm_phases_ms']['reduce_ms'], d['msm_phases_ms']['sort_ms']))"
done
done
