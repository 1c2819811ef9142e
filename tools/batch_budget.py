#!/usr/bin/env python3
"""The batch's issue budget per proof (tools/prof_batch_r05.sh).

    python tools/batch_budget.py <pmc dir A> <pmc dir B> <line of run A> <line of run B> <serial proof budget json> <batch line json> <out json>

Two --pmc passes (SQ_INSTS_VALU, SQ_WAVES) of the same batch command that differ only in the number of proofs: per kernel name
(sum over B) - (sum over A) = the instructions of that many proofs in the batch's own shape.  Each kernel's instructions are priced
with the busy time per instruction the serial proof budget measured for the same kernel (duration x VALUBusy / SQ_INSTS_VALU):
sum = the batch's issue floor per proof; efficiency = floor x proofs/s of the plain run."""
import collections
import json
import sys

from proof_budget import pmc_launches


def totals(directory):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for d in pmc_launches(directory) or []:
        cnt[d["name"]] += 1
        for c, v in d["c"].items():
            agg[d["name"]][c] += v
    return agg, cnt


def main():
    dir_a, dir_b, line_a, line_b, serial_path, line_path, out_path = sys.argv[1:8]
    last_line = lambda path: json.loads(open(path).read().strip().splitlines()[-1])
    # proofs the two profiled processes made, warm-up and pre-sweep included (`proofs_made_in_run` of their own lines)
    dproofs = last_line(line_b)["proofs_made_in_run"] - last_line(line_a)["proofs_made_in_run"]
    (a, ca), (b, cb) = totals(dir_a), totals(dir_b)
    serial = json.load(open(serial_path))
    line = json.loads(open(line_path).read().strip().splitlines()[-1])
    price = {}        # us of VALU-busy time per wave-instruction, per kernel, from the serial proof
    for k in serial["kernels"]:
        if k.get("SQ_INSTS_VALU") and "issue_floor_us" in k:
            price[k["kernel"]] = k["issue_floor_us"] / k["SQ_INSTS_VALU"]
    mean_price = serial["issue_floor_ms"] * 1e3 / serial["valu_wave_instructions_total"]
    rows, floor_us, insts_total, unpriced = [], 0.0, 0.0, 0.0
    for name in sorted(b, key=lambda n_: -(b[n_]["SQ_INSTS_VALU"] - a[n_]["SQ_INSTS_VALU"] if n_ in a else b[n_]["SQ_INSTS_VALU"])):
        di = (b[name]["SQ_INSTS_VALU"] - a[name]["SQ_INSTS_VALU"]) / dproofs if name in a else b[name]["SQ_INSTS_VALU"] / dproofs
        dl = (cb[name] - ca[name]) / dproofs
        if di <= 0 and dl <= 0:
            continue
        p = price.get(name)
        if p is None:
            unpriced += di
        us = di * (p if p is not None else mean_price)
        floor_us += us
        insts_total += di
        serial_k = next((k for k in serial["kernels"] if k["kernel"] == name), None)
        rows.append({"kernel": name, "launches_per_proof": round(dl, 3), "valu_wave_instructions_per_proof": round(di),
                     "serial_proof_instructions": serial_k.get("SQ_INSTS_VALU") if serial_k else None,
                     "issue_us_per_proof": round(us, 1), "priced_by": "own kernel (serial budget)" if p is not None else "mean of the proof"})
    pps = line.get("proofs_per_s")
    out = {"what": "the 1024-proof k = 17 batch, per proof: VALU wave-instructions by kernel from the difference of two --pmc passes "
                   f"({dproofs} proofs apart), priced with the serial proof's measured VALU-busy time per instruction of the same kernel",
           "valu_wave_instructions_per_proof": round(insts_total), "serial_proof_valu_wave_instructions": serial["valu_wave_instructions_total"],
           "launches_per_proof": round(sum(r["launches_per_proof"] for r in rows), 1),
           "issue_floor_ms_per_proof": round(floor_us / 1e3, 3), "serial_proof_issue_floor_ms": serial["issue_floor_ms"],
           "instructions_priced_at_the_mean": round(unpriced),
           "proofs_per_s": pps, "ms_per_proof": round(1e3 / pps, 3) if pps else None,
           "efficiency_issue_floor_over_time_per_proof": round(floor_us / 1e3 * pps / 1e3, 3) if pps else None,
           "kernels": rows}
    json.dump(out, open(out_path, "w"), indent=1)
    print(f"# batch: {insts_total / 1e6:.0f} M VALU wave-instructions and {out['launches_per_proof']} launches per proof (serial proof: {serial['valu_wave_instructions_total'] / 1e6:.0f} M); "
          f"issue floor {floor_us / 1e3:.3f} ms per proof; {pps} proofs/s = {out['ms_per_proof']} ms per proof; efficiency {out['efficiency_issue_floor_over_time_per_proof']}")
    for r in rows[:30]:
        print(f"  {r['kernel'][:44]:44} {r['launches_per_proof']:7.2f} launches {r['valu_wave_instructions_per_proof'] / 1e6:9.2f} M insts {r['issue_us_per_proof']:8.1f} us  ({r['priced_by']})")


if __name__ == "__main__":
    main()
