#!/bin/bash
# usage (GPU box): acc_trace.sh cfg1 cfg2 ...: per SG_PARAMS setting, one MSM at a time (in-flight 1): when the waves of msm_accumulate leave (msm.acc_trace), and the phase times
set -euo pipefail
for cfg in "$@"; do
  SG_PARAMS="msm.acc_trace=1,$cfg" python bench.py --no-extras --no-cpu --steps 6 --warmup 2 --in-flight 1 > /tmp/acc_trace.json 2> /tmp/acc_trace.err
  grep acc_trace /tmp/acc_trace.err | tail -1
  SG_PARAMS="$cfg" python bench.py --no-extras --no-cpu --steps 20 --warmup 3 --in-flight 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg  step(in-flight 1) %.3f ms  accumulate %.3f ms  reduce %.3f ms  sort %.3f' % (d['ms_per_step'], d['msm_phases_ms']['accumulate_ms'], d['msm_phases_ms']['reduce_ms'], d['msm_phases_ms']['sort_ms']))"
done
