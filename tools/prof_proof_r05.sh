#!/bin/bash
# Round 5: a counter-backed budget under ONE k = 17 proof (tools/create_proof_cpp with SG_PROVER_SERIAL=1: every kernel of a
# proof alone on the device) -- the kernel trace for the durations and SEPARATE --pmc passes (never combined with a trace
# domain) for instruction counts, VALU busy / lane utilisation, FETCH_SIZE, WRITE_SIZE and the wave-cycle split.
# tools/proof_budget.py joins them launch by launch (the program is deterministic: the same launch sequence in every pass)
# into gpurun_out/<tag>_profiles/<tag>_proof_budget.json.
# usage (GPU box, from the repo root): TAG=r05a tools/prof_proof_r05.sh
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag="${TAG:-r05a}"
work="$GRAFT_REPO_ROOT/gpurun_out/${tag}_work"
out="$GRAFT_REPO_ROOT/gpurun_out/${tag}_profiles"
mkdir -p "$work" "$out"
(rocprofv3 -L > "$out/counters_available.txt" 2>&1) || true
python - "$work/bundle17.bin" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from bench import snapshot_tree
from circuits_halo2_amd import api, ffi, prover
ffi.check(ffi.lib().sg_init(0))
tree = snapshot_tree(20, 2)
params, pk, vk = api.generate_setup_artifacts(17, None, api.MstInclusionCircuit.init_empty(20, 2, 8))
c = api.MstInclusionCircuit.init_from_tree(tree, 5)
adv = api._advice_columns(pk, c)
prover.export_bundle(sys.argv[1], params, pk, adv, c.instances()[0])
print("bundle written")
PY
CP="$GRAFT_REPO_ROOT/tools/create_proof_cpp"
bundle="$work/bundle17.bin"
# without the profiler: the wall clock of the compiled driver, best of 30 (real schedule), and of the serial schedule
"$CP" "$bundle" "$work/proof.bin" 30 > "$out/${tag}_create_proof_cpp.json"
cat "$out/${tag}_create_proof_cpp.json"
SG_PROVER_SERIAL=1 "$CP" "$bundle" "$work/proof.bin" 30 > "$out/${tag}_create_proof_cpp_serial.json"
cat "$out/${tag}_create_proof_cpp_serial.json"
rm -rf "$work"/serial_*
(cd /tmp && SG_PROVER_SERIAL=1 SG_ACC_LOG="$work/serial_trace_acclog.json" rocprofv3 --kernel-trace --stats -d "$work/serial_trace" -- "$CP" "$bundle" "$work/proof.bin" 8 > "$work/cpp_trace.json" 2> "$work/rocprof_trace.err")
echo "trace done"
pass() {   # pass <name> <counters...>: one --pmc pass of the same program; a pass that fails (unknown counter) is reported, not fatal
  local name="$1"; shift
  (cd /tmp && SG_PROVER_SERIAL=1 SG_ACC_LOG="$work/serial_${name}_acclog.json" rocprofv3 --pmc "$@" -d "$work/serial_$name" -- "$CP" "$bundle" "$work/proof.bin" 4 > "$work/cpp_$name.json" 2> "$work/rocprof_$name.err") \
    && echo "pass $name done" || { echo "pass $name FAILED"; tail -5 "$work/rocprof_$name.err"; }
}
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass valu VALUBusy VALUUtilization
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
python tools/proof_kernels.py "$work/serial_trace" > "$out/${tag}_proof_kernels_serial.txt"
python tools/proof_budget.py "$work" "$tag" "$out/${tag}_proof_budget.json" | tee "$out/${tag}_proof_budget.txt"
rm -rf "$bundle" "$work"/serial_*/
ls "$out"
