"""Restated verifier of `MstInclusionCircuit<4, 2, 8>` proofs (EVM / Keccak flavour), with Python integers.

TEST INFRASTRUCTURE ONLY (see oracle/pyref.py).  This is the CPU restatement of the *caller side* of the hot
path -- SURVEY.md §8(f) rows 1 and 3: the constraint system whose quotient `evaluate_h` computes, the
Keccak transcript (K7), the Lagrange / instance evaluations and the SHPLONK (BDFG21) opening check.  It follows
the reference's generated verifier, cited per function:

  transcript ....... contracts/src/InclusionVerifier.sol:85-110, 281-367
  Lagrange evals ... :423-492
  gates ............ :495-909   (Poseidon Pow5 chips of width 2, merkle-sum-tree chips, range-check lookup;
                                 circuit source: zk_prover/src/circuits/merkle_sum_tree.rs:143-196,
                                 chips/poseidon/poseidon_chip.rs, chips/merkle_sum_tree.rs, chips/range/range_check.rs)
  permutation ...... :910-951   (6 columns in chunks of 4: two grand products)
  lookup ........... :952-992
  quotient commit .. :1004-1023
  SHPLONK .......... :1025-1363, pairing :1395-1402

Pinned two ways (tests/test_verifier_cpu.py): (1) it accepts the reference's own shipped proof
zk_prover/examples/inclusion_proof_solidity_calldata.json (K6) and rejects it after any single-byte change;
(2) every intermediate value (the eight challenges, l_0 / l_last / l_blind, instance evaluation, quotient
evaluation, quotient commitment, r_eval, both pairing inputs) equals the value the reference's verifier itself
computes, recorded in tests/golden/k6_verifier_trace.json by oracle/yul_verifier_run.py.

The gate set is written from the circuit's structure (round constants in fixed columns, MDS / MDS^-1 from the
Poseidon parameters, the compressed selector of the simple selectors), not as a flat list of products.
"""
from __future__ import annotations

from .pairing import pairing_check
from .pyref import Q, R, DELTA, g1_add, g1_mul, keccak256, omega_for
from functools import lru_cache

from .poseidon_params import generate as _generate_poseidon


@lru_cache(maxsize=None)
def poseidon_generate():
    return _generate_poseidon()


K = 11
NUM_ADVICE, NUM_FIXED, NUM_INSTANCE_COLS = 3, 11, 1
BLINDING_FACTORS = 5
ROT_LAST = -(BLINDING_FACTORS + 1)
# columns of the permutation argument, in the order of their sigma polynomials
PERMUTATION_COLUMNS = [("f", 2), ("a", 0), ("a", 1), ("f", 3), ("a", 2), ("i", 0)]
PERMUTATION_CHUNK = 4  # degree 6 -> chunks of degree - 2 columns

# order in which the prover writes the 35 evaluations at x (proof bytes 0x380 ..), :500-1000 `calldataload` slots
EVAL_ORDER = ([("a", 0, 0), ("a", 1, 0), ("a", 0, 1), ("a", 1, 1), ("a", 2, 0), ("a", 1, -1), ("a", 0, -1),
               ("f", 2, 0), ("f", 3, 0), ("f", 0, 0), ("f", 1, 0)] + [("f", j, 0) for j in range(4, 11)] +
              [("random", 0, 0)] + [("sigma", j, 0) for j in range(6)] +
              [("z", 0, 0), ("z", 0, 1), ("z", 0, ROT_LAST), ("z", 1, 0), ("z", 1, 1),
               ("lz", 0, 0), ("lz", 0, 1), ("pin", 0, 0), ("pin", 0, -1), ("ptab", 0, 0)])
# commitments in proof order (phase by phase), :301-345
COMMIT_ORDER = [("a", 0), ("a", 1), ("a", 2), ("pin", 0), ("ptab", 0), ("z", 0), ("z", 1), ("lz", 0), ("random", 0)] + \
               [("h", j) for j in range(5)]
# SHPLONK rotation sets in nu order; inside a set the polynomials in increasing power of zeta (:1159-1232, 1280-1340)
ROTATION_SETS = [
    ((-1, 0, 1), [("a", 0), ("a", 1)]),
    ((0,), [("a", 2), ("ptab", 0), ("f", 2), ("f", 3), ("f", 0), ("f", 1)] + [("f", j) for j in range(4, 11)] +
           [("sigma", j) for j in range(6)] + [("h", None), ("random", 0)]),
    ((ROT_LAST, 0, 1), [("z", 0)]),
    ((0, 1), [("z", 1), ("lz", 0)]),
    ((-1, 0), [("pin", 0)]),
]


def inv(a):
    return pow(a, -1, R)


# ------------------------------------------------------------------ transcript (K7)
class EvmTranscript:
    """Keccak256 transcript of halo2_solidity_verifier: the buffer starts with the vk digest; a challenge is
    keccak(buffer) mod r and the 32-byte hash becomes the new buffer; a second challenge without new input
    hashes `hash || 0x01` (:85-110)."""

    def __init__(self, vk_digest: int):
        self.buf = vk_digest.to_bytes(32, "big")

    def absorb_scalar(self, v: int):
        self.buf += v.to_bytes(32, "big")

    def absorb_point(self, p):
        self.buf += p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")

    def squeeze(self) -> int:
        h = keccak256(self.buf)
        self.buf = h
        return int.from_bytes(h, "big") % R

    def squeeze_again(self) -> int:
        h = keccak256(self.buf[:32] + b"\x01")
        self.buf = h
        return int.from_bytes(h, "big") % R


class Blake2bTranscript:
    """`Blake2bRead<_, G1Affine, Challenge255<_>>` as `full_verifier` uses it [REF zk_prover/src/circuits/utils.rs:118]
    (halo2_proofs transcript.rs, restated from its published source -- SURVEY.md Appendix A): Blake2b-512 personalised
    "Halo2-Transcript" over  prefix 1 || x || y  per point,  prefix 2 || repr  per scalar (32-byte little-endian
    canonical values), prefix 0 per challenge; a challenge is the digest of the stream so far as a 512-bit
    little-endian integer mod r.  Uses this directory's own Blake2b (pyref.blake2b, pinned on RFC 7693)."""

    def __init__(self, vk_digest: int):
        self.stream = b""
        self.absorb_scalar(vk_digest)

    def absorb_scalar(self, v: int):
        self.stream += b"\x02" + v.to_bytes(32, "little")

    def absorb_point(self, p):
        self.stream += b"\x01" + p[0].to_bytes(32, "little") + p[1].to_bytes(32, "little")

    def squeeze(self) -> int:
        from .pyref import blake2b
        self.stream += b"\x00"
        return int.from_bytes(blake2b(self.stream, 64, b"Halo2-Transcript"), "little") % R

    squeeze_again = squeeze


def decompress_g1(enc: bytes):
    """halo2curves `G1Affine::from_bytes`: x little-endian, bit 6 of the last byte = parity of y, bit 7 = infinity
    (recalled layout, see circuits_halo2_amd/prover.py::compress_g1); raises ValueError for a non-point"""
    if len(enc) != 32:
        raise ValueError("compressed point length")
    last = enc[31]
    x = int.from_bytes(enc[:31] + bytes([last & 0x3F]), "little")
    if last & 0x80:
        if x or last & 0x40:
            raise ValueError("non-canonical identity")
        raise ValueError("point at infinity in a proof")
    if x >= Q:
        raise ValueError("x not reduced")
    y = pow((x * x * x + 3) % Q, (Q + 1) // 4, Q)       # q = 3 mod 4
    if (y * y - x * x * x - 3) % Q:
        raise ValueError("not on the curve")
    if (y & 1) != ((last >> 6) & 1):
        y = Q - y
    return (x, y)


def parse_proof_blake2b(proof: bytes):
    """1632 bytes: the same sequence as parse_proof with compressed points and little-endian scalars"""
    if len(proof) != 32 * (len(COMMIT_ORDER) + len(EVAL_ORDER) + 2):
        raise ValueError("proof length")
    pos = 0
    comms = {}
    for key in COMMIT_ORDER:
        comms[key] = decompress_g1(proof[pos:pos + 32])
        pos += 32
    evals = {}
    for key in EVAL_ORDER:
        evals[key] = int.from_bytes(proof[pos:pos + 32], "little")
        pos += 32
    return comms, evals, decompress_g1(proof[pos:pos + 32]), decompress_g1(proof[pos + 32:pos + 64])


def parse_proof(proof: bytes):
    """2144 bytes: 14 commitments (x || y big-endian), 35 evaluations, W, W' (SURVEY.md appendix C)."""
    if len(proof) != 64 * len(COMMIT_ORDER) + 32 * len(EVAL_ORDER) + 128:
        raise ValueError("proof length")
    word = lambda o: int.from_bytes(proof[o:o + 32], "big")
    pos = 0
    comms = {}
    for key in COMMIT_ORDER:
        comms[key] = (word(pos), word(pos + 32))
        pos += 64
    evals = {}
    for key in EVAL_ORDER:
        evals[key] = word(pos)
        pos += 32
    w = (word(pos), word(pos + 32))
    w2 = (word(pos + 64), word(pos + 96))
    return comms, evals, w, w2


def on_curve(p) -> bool:
    x, y = p
    return x < Q and y < Q and (y * y - x * x * x - 3) % Q == 0


# ------------------------------------------------------------------ constraint system
def gate_values(q, n_currencies: int = 2):
    """The 19 gate polynomials of the circuit at one point; `q(kind, column, rotation)` returns the value of a
    column there (an evaluation at x * omega^rotation for the verifier, a cell of the row for a row-wise check).

    Fixed columns: f0, f1 = rc_a; f2, f3 = rc_b (round constants of the two Pow5 chips, which share columns);
    f4 = the 8-bit range table; f5 = the lookup's selector; f6 = the compressed simple selector of the four
    merkle-sum-tree / pad-and-add gates (value v in 1..4 enables the gate whose product omits (v - f6));
    f7 / f8 = s_full / s_partial of the first Poseidon chip, f9 / f10 of the second."""
    _, mds, mds_inv = poseidon_generate()
    a = lambda c, r=0: q("a", c, r)
    f = lambda c: q("f", c, 0)
    out = []

    def pow5(v):
        v2 = v * v % R
        return v2 * v2 % R * v % R

    def poseidon_chip(s_full, s_partial):
        # full round: next_i = sum_j mds[i][j] * (cur_j + rc_a_j)^5
        sbox = [pow5((a(j) + f(j)) % R) for j in range(2)]
        for i in range(2):
            out.append(s_full * ((mds[i][0] * sbox[0] + mds[i][1] * sbox[1] - a(i, 1)) % R) % R)
        # two partial rounds per row; a2 holds the s-box output of the first one
        out.append(s_partial * ((sbox[0] - a(2)) % R) % R)
        mid = [a(2), (a(1) + f(1)) % R]
        r_mid = [(mds[i][0] * mid[0] + mds[i][1] * mid[1]) % R for i in range(2)]
        nxt = [(mds_inv[i][0] * a(0, 1) + mds_inv[i][1] * a(1, 1)) % R for i in range(2)]
        out.append(s_partial * ((pow5((r_mid[0] + f(2)) % R) - nxt[0]) % R) % R)
        out.append(s_partial * ((r_mid[1] + f(3) - nxt[1]) % R) % R)

    def simple_selector(value):
        # f6 * prod_{v != value} (v - f6): non-zero only on rows where f6 == value
        s = f(6)
        for v in range(1, 5):
            if v != value:
                s = s * ((v - f(6)) % R) % R
        return s

    def pad_and_add(value):
        s = simple_selector(value)
        out.append(s * ((a(0, -1) + a(0) - a(0, 1)) % R) % R)
        out.append(s * ((a(1, -1) - a(1, 1)) % R) % R)

    poseidon_chip(f(7), f(8))
    pad_and_add(3)
    poseidon_chip(f(9), f(10))
    pad_and_add(4)
    # swap gate (chips/merkle_sum_tree.rs): a2 is the swap bit
    s = simple_selector(1)
    out.append(s * a(2) % R * ((1 - a(2)) % R) % R)
    out.append(s * (((a(1) - a(0)) * a(2) + a(0) - a(0, 1)) % R) % R)
    out.append(s * (((a(0) - a(1)) * a(2) + a(1) - a(1, 1)) % R) % R)
    # sum gate, once per currency
    s = simple_selector(2)
    for _ in range(n_currencies):
        out.append(s * ((a(0) + a(1) - a(2)) % R) % R)
    return out


def lookup_input_table(q):
    """range check of 8 bits: f5 * (a0 - 2^8 * a0_next) must be in the table column f4 (:953-969)"""
    return q("f", 5, 0) * ((q("a", 0, 0) - 256 * q("a", 0, 1)) % R) % R, q("f", 4, 0)


def quotient_numerator(q, ch, lag, n_currencies: int = 2):
    """All constraints folded with powers of y (Horner, first gate highest), :495-1000.
    ch: challenges theta beta gamma y x;  lag: l_0, l_last, l_blind, instance_eval."""
    beta, gamma, y, x = ch["beta"], ch["gamma"], ch["y"], ch["x"]
    l_0, l_last, l_blind = lag["l_0"], lag["l_last"], lag["l_blind"]
    active = (1 - l_last - l_blind) % R
    terms = list(gate_values(q, n_currencies))
    # permutation argument
    chunks = [PERMUTATION_COLUMNS[i:i + PERMUTATION_CHUNK] for i in range(0, len(PERMUTATION_COLUMNS), PERMUTATION_CHUNK)]
    z = lambda j, r=0: q("z", j, r)
    terms.append(l_0 * ((1 - z(0)) % R) % R)
    terms.append(l_last * ((z(len(chunks) - 1) ** 2 - z(len(chunks) - 1)) % R) % R)
    for j in range(1, len(chunks)):
        terms.append(l_0 * ((z(j) - z(j - 1, ROT_LAST)) % R) % R)
    shift = beta * x % R  # beta * delta^i * x for the i-th permutation column
    col = 0
    for j, chunk in enumerate(chunks):
        lhs, rhs = z(j, 1), z(j)
        for kind, c in chunk:
            v = lag["instance_eval"] if kind == "i" else q(kind, c, 0)
            lhs = lhs * ((v + beta * q("sigma", col, 0) + gamma) % R) % R
            rhs = rhs * ((v + shift + gamma) % R) % R
            shift = shift * DELTA % R
            col += 1
        terms.append((lhs - rhs) * active % R)
    # lookup argument
    inp, tab = lookup_input_table(q)
    lz, pin, ptab = (lambda r=0: q("lz", 0, r)), (lambda r=0: q("pin", 0, r)), q("ptab", 0, 0)
    terms.append(l_0 * ((1 - lz()) % R) % R)
    terms.append(l_last * ((lz() ** 2 - lz()) % R) % R)
    lhs = lz(1) * ((pin() + beta) % R) % R * ((ptab + gamma) % R) % R
    rhs = lz() * ((inp + beta) % R) % R * ((tab + gamma) % R) % R
    terms.append(active * ((lhs - rhs) % R) % R)
    terms.append(l_0 * ((pin() - ptab) % R) % R)
    terms.append(active * ((pin() - ptab) % R) % R * ((pin() - pin(-1)) % R) % R)
    acc = 0
    for t in terms:
        acc = (acc * y + t) % R
    return acc


def lagrange_evaluations(x, instances, k=K):
    """l_last (row -6), l_blind (rows -5 .. -1), l_0 and the instance column's evaluation at x (:423-492)."""
    n = 1 << k
    omega = omega_for(k)
    x_n = pow(x, n, R)
    common = (x_n - 1) * inv(n) % R
    l = lambda i: common * pow(omega, i, R) % R * inv((x - pow(omega, i, R)) % R) % R
    return {"x_n": x_n, "l_last": l(ROT_LAST), "l_blind": sum(l(i) for i in range(ROT_LAST + 1, 0)) % R, "l_0": l(0),
            "instance_eval": sum(l(i) * v for i, v in enumerate(instances)) % R}


# ------------------------------------------------------------------ SHPLONK
def shplonk_pairing_inputs(comms, evals_of, x, zeta, nu, mu, w, w2, k=K):
    """-> (lhs, rhs) with e(lhs, [1]_2) == e(rhs, [s]_2)  iff all openings hold.
    comms[key] -> commitment; evals_of(key, rotation) -> claimed evaluation of that polynomial at x * omega^rotation.
    Everything is divided by Z_{T \\ S_0}(mu) as the reference does (:1047-1070, 1146-1158)."""
    omega = omega_for(k)
    point = lambda rot: x * pow(omega, rot % (1 << k), R) % R
    all_rots = sorted({r for rots, _ in ROTATION_SETS for r in rots})
    mu_minus = {r: (mu - point(r)) % R for r in all_rots}
    diffs = []
    for rots, _ in ROTATION_SETS:
        d = 1
        for r in all_rots:
            if r not in rots:
                d = d * mu_minus[r] % R
        diffs.append(d)
    d0_inv = inv(diffs[0])
    diffs = [d * d0_inv % R for d in diffs]
    z_s0 = 1
    for r in ROTATION_SETS[0][0]:
        z_s0 = z_s0 * mu_minus[r] % R
    lhs, r_eval, nu_pow = None, 0, 1
    for (rots, polys), diff in zip(ROTATION_SETS, diffs):
        # barycentric weights of the set's points at mu
        wts = []
        for r in rots:
            d = mu_minus[r]
            for r2 in rots:
                if r2 != r:
                    d = d * ((point(r) - point(r2)) % R) % R
            wts.append(inv(d))
        norm = inv(sum(wts) % R)
        comm, r_i = None, 0
        for key in reversed(polys):  # Horner in zeta, highest power first
            comm = g1_add(g1_mul(comm, zeta) if comm else None, comms[key])
            at_mu = sum(wt * evals_of(key, r) for wt, r in zip(wts, rots)) % R * norm % R
            r_i = (r_i * zeta + at_mu) % R
        scale = nu_pow * diff % R
        lhs = g1_add(lhs, g1_mul(comm, scale))
        r_eval = (r_eval + scale * r_i) % R
        nu_pow = nu_pow * nu % R
    lhs = g1_add(lhs, g1_mul((1, 2), (-r_eval) % R))
    lhs = g1_add(lhs, g1_mul(w, (-z_s0) % R))
    lhs = g1_add(lhs, g1_mul(w2, mu))
    return lhs, w2, r_eval


def verify(proof: bytes, instances, vk, trace=None, flavour: str = "evm") -> bool:
    """flavour "evm": Keccak transcript, 2144-byte proof (the generated verifier's); "blake2b": `full_verifier`'s
    Blake2b / Challenge255 transcript and compressed 1632-byte proof -- same protocol, same checks.
    vk: {"vk_digest", "fixed_comms" [11], "permutation_comms" [6], "g2", "neg_s_g2"[, "k": 11, "n_currencies": 2]} (integers / int tuples).
    `trace`, if given, is filled with the intermediate values named as in tests/golden/k6_verifier_trace.json."""
    try:
        comms, evals, w, w2 = parse_proof(proof) if flavour == "evm" else parse_proof_blake2b(proof)
    except ValueError:
        return False
    if any(v >= R for v in instances) or any(v >= R for v in evals.values()):
        return False
    if not all(on_curve(p) for p in list(comms.values()) + [w, w2]):
        return False
    t = EvmTranscript(vk["vk_digest"]) if flavour == "evm" else Blake2bTranscript(vk["vk_digest"])
    for v in instances:
        t.absorb_scalar(v)
    ch = {}
    phases = [(3, ["theta"]), (2, ["beta", "gamma"]), (4, ["y"]), (5, ["x"])]
    it = iter(COMMIT_ORDER)
    for count, names in phases:
        for _ in range(count):
            t.absorb_point(comms[next(it)])
        ch[names[0]] = t.squeeze()
        for extra in names[1:]:
            ch[extra] = t.squeeze_again()
    for key in EVAL_ORDER:
        t.absorb_scalar(evals[key])
    ch["zeta"] = t.squeeze()
    ch["nu"] = t.squeeze_again()
    t.absorb_point(w)
    ch["mu"] = t.squeeze()

    x = ch["x"]
    k = vk.get("k", K)
    lag = lagrange_evaluations(x, instances, k)
    q = lambda kind, c, rot: evals[(kind, c, rot)]
    numer = quotient_numerator(q, ch, lag, vk.get("n_currencies", 2))
    quotient_eval = numer * inv((lag["x_n"] - 1) % R) % R
    # h(X) = sum_i x^(n i) h_i(X): one commitment for the five pieces
    h_comm = None
    for j in reversed(range(5)):
        h_comm = g1_add(g1_mul(h_comm, lag["x_n"]) if h_comm else None, comms[("h", j)])
    all_comms = dict(comms)
    all_comms[("h", None)] = h_comm
    for j, c in enumerate(vk["fixed_comms"]):
        all_comms[("f", j)] = tuple(c)
    for j, c in enumerate(vk["permutation_comms"]):
        all_comms[("sigma", j)] = tuple(c)

    def evals_of(key, rot):
        if key == ("h", None):
            return quotient_eval
        return evals[(key[0], key[1], rot)]
    lhs, rhs, r_eval = shplonk_pairing_inputs(all_comms, evals_of, x, ch["zeta"], ch["nu"], ch["mu"], w, w2, k)
    if trace is not None:
        trace.update(ch)
        trace.update({k_: lag[k_] for k_ in ("x_n", "l_last", "l_blind", "l_0", "instance_eval")})
        trace["x_n_minus_1_inv"] = inv((lag["x_n"] - 1) % R)
        trace.update({"quotient_eval": quotient_eval, "quotient_x": h_comm[0], "quotient_y": h_comm[1], "r_eval": r_eval,
                      "pairing_lhs_x": lhs[0], "pairing_lhs_y": lhs[1], "pairing_rhs_x": rhs[0], "pairing_rhs_y": rhs[1]})
    return pairing_check([(lhs, vk["g2"]), (rhs, vk["neg_s_g2"])])
