#!/bin/bash
# Round 5: the 1024-proof batch's instruction budget.  Two --pmc passes (SQ_INSTS_VALU SQ_WAVES; never combined with a trace
# domain) of bench.py --batch-only at 256 and 512 proofs -- everything else identical (set-up, key generation, the pre-sweep at
# --batch-in-flight 64), so the difference is 256 proofs' worth of kernels in the batch's own shape (fused commitment jobs,
# sixteen witnesses per launch) -- and one plain run for the rate.  tools/batch_budget.py prices the instructions with the
# serial proof's measured busy time per instruction (profiles/<tag>_proof_budget.json).
# usage (GPU box, from the repo root): TAG=r05a tools/prof_batch_r05.sh      (after tools/prof_proof_r05.sh of the same tag)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag="${TAG:-r05a}"
work="$GRAFT_REPO_ROOT/gpurun_out/${tag}_work"
out="$GRAFT_REPO_ROOT/gpurun_out/${tag}_profiles"
mkdir -p "$work" "$out"
IN_FLIGHT="${IN_FLIGHT:-64}"
python bench.py --gpus 1 --batch-only --batch-proofs 1024 --batch-repeats 2 --no-cpu --batch-in-flight "$IN_FLIGHT" --wall-limit 300 > "$out/${tag}_batch_line.json" 2> "$work/batch_plain.err"
tail -c 600 "$out/${tag}_batch_line.json"; echo
for total in 256 512; do
  rm -rf "$work/batch_insts_$total"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d "$work/batch_insts_$total" -- python3 bench.py --gpus 1 --batch-only --batch-proofs "$total" --batch-repeats 1 --no-cpu \
      --batch-in-flight "$IN_FLIGHT" --wall-limit 500 > "$work/batch_insts_$total.json" 2> "$work/batch_insts_$total.err" \
    && echo "pmc pass $total done" || { echo "pmc pass $total FAILED"; tail -5 "$work/batch_insts_$total.err"; }
done
python tools/batch_budget.py "$work/batch_insts_256" "$work/batch_insts_512" "$work/batch_insts_256.json" "$work/batch_insts_512.json" "$out/${tag}_proof_budget.json" "$out/${tag}_batch_line.json" "$out/${tag}_batch_budget.json" | tee "$out/${tag}_batch_budget.txt"
rm -rf "$work"/batch_insts_*/
