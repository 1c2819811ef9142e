#!/bin/bash
# wave priority 3 for every kernel but msm_accumulate (side_prio) x the accumulation's waves per SIMD x steps in flight: headline, proof, batch
set -euo pipefail
for rep in 1 2; do
for cfg in "side_prio=0,msm.acc_waves=3,msm.red_lean=0" "side_prio=1" "side_prio=1,msm.acc_waves=3,msm.red_lean=0" "side_prio=1,msm.red_lean=0"; do
  for f in 3 4 6; do
    SG_PARAMS=$cfg python bench.py --no-extras --no-cpu --steps 60 --in-flight $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg in_flight=$f  %.1f M points/s  step %.3f ms  sequential %.3f ms  accumulate %.3f ms  reduce %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['sequential']['ms_per_step'], d['msm_phases_ms']['accumulate_ms'], d['msm_phases_ms']['reduce_ms']))"
  done
done
done
bash tools/ab_proof_knobs.sh SG_PARAMS=side_prio=0 SG_PARAMS=side_prio=0,msm.acc_waves_fixed=3 SG_PARAMS=msm.acc_waves_fixed=3 SG_PARAMS=side_prio=0 SG_PARAMS=side_prio=0,msm.acc_waves_fixed=3 SG_PARAMS=msm.acc_waves_fixed=3 2>&1 | grep -v amdgpu
for cfg in side_prio=1 side_prio=0 side_prio=1,msm.acc_waves_fixed=3 side_prio=0,msm.acc_waves_fixed=3 side_prio=1 side_prio=0; do echo -n "$cfg  "; SG_PARAMS=$cfg python tools/run_batch.py 16 768 1 2>&1 | tail -1; done
