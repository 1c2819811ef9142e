#!/usr/bin/env python3
"""Regenerates tests/golden/* (run in the authoring container only; needs /root/reference).

Nothing of the reference's Rust is executed (no toolchain exists for it).
Fixtures are DATA:
  hermez-raw-11      byte copy of the reference's SRS test fixture backend/ptau/hermez-raw-11
                     (used by backend/src/tests.rs:238-242); pins K1/K3.
  kat.json           constants parsed from contracts/src/InclusionVerifier.sol:217-271
                     (vk_digest, k, n_inv, omega, omega_inv, fixed_comms, permutation_comms;
                     K2 = fixed_comms[4], K4) + small vectors computed by the independent
                     big-integer twin oracle/pyref.py (NTT 2^4, coeff_to_extended 2^4->2^7,
                     MSM answers as f(tau)*G).
  ntt_k11_{in,out}.bin, msm_tau_k10_{scalars,bases}.bin   pyref-generated vectors.
  entry_*.csv        byte copies of the reference's CSV data fixtures csv/*.csv (the inputs of its Merkle-sum-tree and
                     circuit tests: 13 / 16 / 17 entries, switched order, one modified entry, big integers, an overflowing balance).
"""
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyref as P  # noqa: E402

REF = "/root/reference"


CSVS = ["entry_16.csv", "entry_13.csv", "entry_17.csv", "entry_16_switched_order.csv", "entry_16_modified.csv",
        "entry_16_bigints.csv", "entry_16_overflow.csv"]


def copy_csvs():
    """the reference's CSV data fixtures (csv/*.csv: the inputs of zk_prover/src/merkle_sum_tree/tests.rs and
    zk_prover/src/circuits/tests.rs), byte copies"""
    for name in CSVS:
        shutil.copyfile(os.path.join(REF, "csv", name), os.path.join(HERE, name))
        os.chmod(os.path.join(HERE, name), 0o644)


def main():
    shutil.copyfile(os.path.join(REF, "backend/ptau/hermez-raw-11"), os.path.join(HERE, "hermez-raw-11"))
    os.chmod(os.path.join(HERE, "hermez-raw-11"), 0o644)
    sol = open(os.path.join(REF, "contracts/src/InclusionVerifier.sol")).read()
    kat = {"source": "contracts/src/InclusionVerifier.sol:217-271"}
    for name in ("vk_digest", "k", "n_inv", "omega", "omega_inv", "omega_inv_to_l"):
        m = re.search(r"mstore\(0x[0-9a-f]+, (0x[0-9a-f]{64})\) // %s\b" % name, sol)
        kat[name] = m.group(1)
    for fam in ("fixed_comms", "permutation_comms"):
        pts = {}
        for m in re.finditer(r"mstore\(0x[0-9a-f]+, (0x[0-9a-f]{64})\) // %s\[(\d+)\]\.([xy])" % fam, sol):
            pts.setdefault(int(m.group(2)), {})[m.group(3)] = m.group(1)
        kat[fam] = [[pts[i]["x"], pts[i]["y"]] for i in range(len(pts))]
    m = re.search(r"let delta := (\d+)", sol)
    kat["delta"] = m.group(1) if m else str(P.DELTA)
    kat["root_of_unity_2_28"] = hex(P.ROOT_OF_UNITY)
    kat["zeta"] = hex(P.ZETA)

    # pyref-generated small vectors -----------------------------------------------------
    a4 = P.random_fr(P.DEFAULT_SEED + 4, 16)
    kat["ntt_k4"] = {"in": [hex(x) for x in a4], "out": [hex(x) for x in P.dft_naive(a4, P.omega_for(4))]}
    kat["coeff_to_extended_k4_e7"] = {"in": [hex(x) for x in a4],
                                      "out": [hex(x) for x in P.coeff_to_extended(a4, 4, 7)]}
    kat["t_evaluations_k4_e7"] = [hex(x) for x in P.t_evaluations(4, 7)]
    a11 = P.random_fr(P.DEFAULT_SEED + 11, 2048)
    open(os.path.join(HERE, "ntt_k11_in.bin"), "wb").write(P.frs_to_bytes(a11))
    open(os.path.join(HERE, "ntt_k11_out.bin"), "wb").write(P.frs_to_bytes(P.ntt(a11, P.omega_for(11), 11)))

    # synthetic "unsafe setup" SRS g[i] = tau^i G (ParamsKZG::setup, utils.rs:70), n = 2^10
    tau = P.random_fr(0x7A55, 1)[0]
    n = 1024
    sc = P.random_fr(P.DEFAULT_SEED, n)
    bases, cur = [], 1
    for _ in range(n):
        bases.append(P.g1_mul(P.G1_GEN, cur))
        cur = cur * tau % P.R
    f_tau = sum(s * pow(tau, i, P.R) for i, s in enumerate(sc)) % P.R
    ans = P.g1_mul(P.G1_GEN, f_tau)
    open(os.path.join(HERE, "msm_tau_k10_scalars.bin"), "wb").write(P.frs_to_bytes(sc))
    open(os.path.join(HERE, "msm_tau_k10_bases.bin"), "wb").write(b"".join(P.g1_to_bytes(b) for b in bases))
    kat["msm_tau_k10"] = {"tau": hex(tau), "seed": hex(P.DEFAULT_SEED), "answer": [hex(ans[0]), hex(ans[1])]}
    # witness-like sparse scalars: 97% zero, rest < 2^64 / bytes (advice-column shape)
    sp = [0] * n
    rnd = P.splitmix64_stream(99, 2 * n)
    for i in range(n):
        if rnd[i] % 32 == 0:
            sp[i] = rnd[n + i] if rnd[i] % 64 else rnd[n + i] & 0xFF
    f_tau = sum(s * pow(tau, i, P.R) for i, s in enumerate(sp)) % P.R
    ans = P.g1_mul(P.G1_GEN, f_tau)
    kat["msm_tau_k10_sparse"] = {"scalars": [hex(x) for x in sp], "answer": [hex(ans[0]), hex(ans[1])] if ans else None}
    # --- witness side (K5): the reference's own data files and the constants its tests pin
    copy_csvs()
    tests_rs = open(os.path.join(REF, "zk_prover/src/circuits/tests.rs")).read()
    backend_rs = open(os.path.join(REF, "backend/src/tests.rs")).read()
    leafs = re.findall(r'"(0x[0-9a-f]{64})"', tests_rs)
    kat["k5"] = {"source": "zk_prover/src/circuits/tests.rs:341,346; backend/src/tests.rs:265,268; "
                           "zk_prover/src/merkle_sum_tree/tests.rs:24",
                 "leaf0": leafs[0], "leaf1": leafs[1],
                 "root": re.search(r'mst_root: "(0x[0-9a-f]{64})"', backend_rs).group(1),
                 "root_balances": [556862, 556862]}
    # the regenerated Poseidon parameters must equal the 136 constants of the reference's
    # chips/poseidon/poseidon_params.rs (values only are compared; nothing is copied)
    from oracle import poseidon_params
    rcs, mds, inv = poseidon_params.generate()
    src = open(os.path.join(REF, "zk_prover/src/chips/poseidon/poseidon_params.rs")).read()
    vals = []
    for m in re.finditer(r"Fp::from_raw\(\[\s*(0x[0-9a-f_]+),\s*(0x[0-9a-f_]+),\s*(0x[0-9a-f_]+),\s*(0x[0-9a-f_]+),?\s*\]\)", src):
        l = [int(x.replace("_", ""), 16) for x in m.groups()]
        vals.append(l[0] | l[1] << 64 | l[2] << 128 | l[3] << 192)
    assert vals == [x for r in rcs for x in r] + [x for r in mds for x in r] + [x for r in inv for x in r]
    import hashlib
    kat["poseidon_t2_sha256"] = hashlib.sha256(repr((rcs, mds)).encode()).hexdigest()
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
