#!/bin/bash
# The 1024-proof batch (bench.py --batch-only) under the two ways a host thread can wait for the device
# (SUMMA_WAIT_SLEEP_US=N for the length of the batch, 0 = the runtime's polling wait; csrc/host_wait.h, batch.prove_batch) and under CPU shares of 2 / 4 / all cores: proofs/s, CPU ms per proof.
# usage (GPU box): tools/batch_wait_sweep.sh <tag>
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
tag="${1:-r04}"
out="gpurun_out/${tag}_batch_wait_sweep.txt"
: > "$out"
for share in 2 4 0; do
  for nap in 0 25 100; do
    for inflight in 16 32; do
      line=$(SUMMA_WAIT_SLEEP_US=$nap python bench.py --gpus 1 --batch-only --cpu-share $share --batch-proofs 1024 --batch-repeats 1 --no-cpu \
             --batch-in-flight $inflight --wall-limit 200 2>/dev/null | tail -1)
      python - "$share" "$nap" "$inflight" "$line" >> "$out" <<'PY'
import json, sys
share, nap, inflight, line = sys.argv[1:5]
d = json.loads(line)
print(f"cpu_share {share:>2} (0 = all) | wait_sleep_us asked {nap:>3} used {d.get('wait_sleep_us')} | in_flight asked {inflight:>2} used {d.get('in_flight')} | "
      f"{d.get('proofs_per_s', 0):7.1f} proofs/s | host CPU {d.get('host_cpu_ms_per_proof', 0):6.2f} ms/proof | cores busy {d.get('host_cores_busy_per_gpu', 0):5.2f} | errors {d.get('errors')}")
PY
      tail -1 "$out"
    done
  done
done
