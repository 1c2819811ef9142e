"""The reference's prover API (its layer L3), over the GPU back-end: the functions and the circuit type that the
reference's backend, benches, examples and tests call, with the same names, argument meaning and failure points.

    generate_setup_artifacts(k, params_path, circuit) -> (params, pk, vk)     [REF zk_prover/src/circuits/utils.rs:37-79]
    full_prover(params, pk, circuit, public_inputs) -> proof bytes            [REF :82-107]   Blake2b / Challenge255
    full_verifier(params, vk, proof, public_inputs) -> bool                   [REF :110-131]
    gen_proof_solidity_calldata(params, pk, circuit) -> (proof, instances)    [REF :134-160]  Keccak, re-verified
    field_element_to_solidity_calldata(fe) -> U256                            [REF :199-203]
    MstInclusionCircuit.{init, init_empty, instances, num_instances}          [REF circuits/merkle_sum_tree.rs:31-103]
    ProofSolidityCallData                                                     [REF circuits/types.rs:4-8]

The Rust const generics `<LEVELS, N_CURRENCIES, N_BYTES>` are constructor arguments here.  Everything data-parallel
below this file runs on the device through the C ABI (include/summa_gpu.h); there is no CPU implementation to fall
back to.  Proofs differ run to run (the blinding factors come from the OS entropy source, as the reference's `OsRng`).
"""
from __future__ import annotations

import os
import threading
from dataclasses import dataclass, field

import ctypes as C

import numpy as np

from . import mst_inclusion as M
from . import prover as P
from .merkle_sum_tree import keccak256
from .params import ParamsKZG
from .utils import ints_to_fr

R = M.R
_RINV = pow(1 << 256, -1, R)


def _fr_ints(buf) -> list:
    """Montgomery Fr bytes (numpy uint8 / bytes) -> list of integers"""
    raw = bytes(buf)
    return [int.from_bytes(raw[i:i + 32], "little") * _RINV % R for i in range(0, len(raw), 32)]


@dataclass
class VerifyingKey:
    """what `VerifyingKey<G1Affine>` holds for this circuit: the domain size, the 11 fixed and 6 permutation
    commitments, and `transcript_repr` (the value `vk.hash_into` feeds the transcript; see ProvingKey.vk_digest)"""
    k: int
    n_currencies: int
    fixed_comms: list
    permutation_comms: list
    transcript_repr: int


@dataclass
class ProofSolidityCallData:
    """circuits/types.rs:4-8"""
    proof: str
    public_inputs: list = field(default_factory=list)


class MstInclusionCircuit:
    """`MstInclusionCircuit<LEVELS, N_CURRENCIES, N_BYTES>` [REF circuits/merkle_sum_tree.rs:31-103]: the inclusion
    of an entry (username, balances) in a Merkle sum tree with a given root.  Fields as in the reference: `entry`
    (username field element, balances), `path_indices`, `sibling_leaf_node_hash_preimage` ([username, balances..]),
    `sibling_middle_node_hash_preimages` ([[balances.., left hash, right hash]] per level above the leaves), `root`
    ((hash, balances)); integers throughout."""

    def __init__(self, levels: int, n_currencies: int, n_bytes: int, entry, path_indices, sibling_leaf_node_hash_preimage,
                 sibling_middle_node_hash_preimages, root):
        self.levels, self.n_currencies, self.n_bytes = levels, n_currencies, n_bytes
        self.entry = entry
        self.path_indices = list(path_indices)
        self.sibling_leaf_node_hash_preimage = None if sibling_leaf_node_hash_preimage is None else list(sibling_leaf_node_hash_preimage)
        self.sibling_middle_node_hash_preimages = (None if sibling_middle_node_hash_preimages is None
                                                   else [list(p) for p in sibling_middle_node_hash_preimages])
        self.root = root
        self._assignment = {}
        self._device = None       # (DeviceMerkleSumTree, user index): the witness is synthesized on the device
        self._prefetched = None   # (advice columns, public inputs) laid out ahead of time with other users' (batch.py)

    @classmethod
    def init_empty(cls, levels: int, n_currencies: int = 2, n_bytes: int = 8) -> "MstInclusionCircuit":
        """:75-83 -- the zero entry, zero path and preimages, the empty root: the circuit of key generation"""
        return cls(levels, n_currencies, n_bytes, (0, [0] * n_currencies), [0] * levels, [0] * (n_currencies + 1),
                   [[0] * (n_currencies + 2) for _ in range(levels)], (0, [0] * n_currencies))

    @classmethod
    def init(cls, merkle_proof, levels: int, n_currencies: int | None = None, n_bytes: int = 8) -> "MstInclusionCircuit":
        """:86-103 -- from a `MerkleSumTree.generate_proof(index)` result; the two length assertions are the reference's"""
        name, balances = merkle_proof["entry"]
        nc = len(balances) if n_currencies is None else n_currencies
        assert len(merkle_proof["path_indices"]) == levels
        assert len(merkle_proof["sibling_middle_node_hash_preimages"]) == levels - 1
        assert len(balances) == nc
        # a username is hashed to its field element (entry.rs:21); device snapshots hand the field element over as an int
        username = 0 if name is None else name % R if isinstance(name, int) else int.from_bytes(keccak256(name.encode()), "big") % R
        root_hash, root_bal = merkle_proof["root"]
        return cls(levels, nc, n_bytes, (username, [int(b) for b in balances]), merkle_proof["path_indices"],
                   _fr_ints(merkle_proof["sibling_leaf_node_hash_preimage"]),
                   [_fr_ints(p) for p in merkle_proof["sibling_middle_node_hash_preimages"]],
                   (_fr_ints(root_hash)[0], _fr_ints(root_bal)))

    @classmethod
    def init_from_tree(cls, tree, user_index: int, n_bytes: int = 8) -> "MstInclusionCircuit":
        """`init(tree.generate_proof(user_index))` for a snapshot that lives on the device (DeviceMerkleSumTree): the
        Merkle proof is not brought to the host at all -- the advice columns are laid out by a kernel straight from
        the tree's node arrays (sg_mst_inclusion_witness_dev), only the public inputs come back"""
        if not 0 <= user_index < (1 << tree.depth):
            raise IndexError("Index out of bounds")
        c = cls(tree.depth, tree.n_currencies, n_bytes, None, [(user_index >> l) & 1 for l in range(tree.depth)], None, None, None)
        c._device = (tree, int(user_index))
        return c

    # --- WithInstances [REF circuits/mod.rs:9-12, merkle_sum_tree.rs:47-60]
    def num_instances(self) -> int:
        return 2 + self.n_currencies

    def leaf_hash(self) -> int:
        """`self.entry.compute_leaf().hash`: Poseidon(username, balances..) on the device (sg_mst_leaves_dev)"""
        from .merkle_sum_tree import _hash_batch
        user = ints_to_fr([self.entry[0]])
        bal = ints_to_fr(list(self.entry[1]))
        return _fr_ints(_hash_batch("leaf", user, bal, n=1, nc=self.n_currencies))[0]

    def instances(self) -> list:
        """[[leaf hash, root hash, root balances..]]"""
        if self._prefetched is not None:
            return [list(self._prefetched[1])]
        if self._device is not None:
            return [self._device[0].public_inputs(self._device[1])]
        return [[self.leaf_hash(), self.root[0]] + list(self.root[1])]

    # --- Circuit::synthesize, through halo2's floor planner [REF merkle_sum_tree.rs:228-520]
    def synthesize(self, k: int):
        """the assignment of this circuit over 2^k rows in the reference's own floor plan: fixed columns, permutation,
        advice columns, the values exposed as public inputs (mst_inclusion.reference_assignment).  As halo2's `synthesize`,
        it lays out whatever witness it is given: a tampered entry, path index or an out-of-range balance yields advice
        columns that violate the constraints, and a proof made from them is one the verifier rejects [REF
        circuits/tests.rs:125-152, and the MockProver cases :158-433, restated in tests/test_mock_prover_cpu.py]"""
        if self._device is not None and self.entry is None:   # host view of a device-side circuit: fetch the Merkle proof
            mp = MstInclusionCircuit.init(self._device[0].generate_proof(self._device[1]), self.levels, self.n_currencies, self.n_bytes)
            self.entry, self.root = mp.entry, mp.root
            self.sibling_leaf_node_hash_preimage = mp.sibling_leaf_node_hash_preimage
            self.sibling_middle_node_hash_preimages = mp.sibling_middle_node_hash_preimages
        if k not in self._assignment:
            self._assignment[k] = M.reference_assignment(k, self.entry[0], list(self.entry[1]), self.path_indices,
                                                          self.sibling_leaf_node_hash_preimage,
                                                          self.sibling_middle_node_hash_preimages[:max(0, self.levels - 1)],
                                                          self.n_bytes, lenient=True)
        return self._assignment[k]

    def shape(self):
        return (self.levels, self.n_currencies, self.n_bytes)


def _device_column(ints, n: int):
    """list of n integers (mostly zero beyond the used rows) -> device Montgomery column"""
    import torch
    used = n
    while used and not ints[used - 1]:
        used -= 1
    t = torch.zeros(32 * n, dtype=torch.uint8, device="cuda")
    if used:
        t[:32 * used] = torch.from_numpy(ints_to_fr(ints[:used])).cuda()
    return t


def keygen(params: ParamsKZG, circuit: MstInclusionCircuit, vk_transcript_repr: int | None = None):
    """`keygen_vk` + `keygen_pk` [REF utils.rs:75-76]: synthesize the (empty) circuit for its fixed columns and
    permutation, commit to them (17 MSMs) and transform them into the three bases the prover reads (all on the device).
    The key's digest is halo2's `transcript_repr`, derived (vk_repr.py); `vk_transcript_repr` overrides it for a key
    whose constraint system was built elsewhere."""
    import torch
    from . import ffi
    k = params.k
    n = 1 << k
    # the floor plan (fixed columns, permutation) by the library's compiled replay of `synthesize` (include/summa_circuit.hpp;
    # mst_inclusion.reference_assignment is its Python twin, compared column by column in tests/test_host_logic.py)
    fixed_np = np.empty((M.NUM_FIXED, 32 * n), dtype=np.uint8)
    sigma_np = np.empty((len(M.PERMUTATION_COLUMNS), 32 * n), dtype=np.uint8)
    rows = C.c_uint32(0)
    ffi.check(ffi.lib().sg_mst_inclusion_keygen_columns(C.c_uint32(k), C.c_uint32(circuit.levels), C.c_uint32(circuit.n_currencies),
                                                        C.c_uint32(circuit.n_bytes), ffi.ptr(fixed_np), ffi.ptr(sigma_np), C.byref(rows)))
    pk = P.ProvingKey(params, k, [torch.from_numpy(c).cuda() for c in fixed_np], [torch.from_numpy(c).cuda() for c in sigma_np],
                      circuit.n_currencies)
    pk.circuit_shape = circuit.shape()
    if vk_transcript_repr is not None:
        pk.vk_digest = int(vk_transcript_repr) % R
    vk = VerifyingKey(k, circuit.n_currencies, pk.fixed_comms, pk.permutation_comms, pk.vk_digest)
    pk.vk = vk
    return pk, vk


def generate_setup_artifacts(k: int, params_path: str | None, circuit: MstInclusionCircuit, vk_transcript_repr: int | None = None):
    """[REF utils.rs:37-79]  Load the trusted-setup file (downsizing it to k when it is larger; "k is too large for
    the given params" when it is smaller) or, with no path, run the unsafe setup with a secret from the OS entropy
    source; then generate the verifying and proving keys for `circuit`'s shape.  Returns (params, pk, vk)."""
    if params_path is not None:
        with open(params_path, "rb") as f:                     # "couldn't load params" / "Failed to read params"
            params = ParamsKZG.read(f)
        if params.k < k:
            raise ValueError("k is too large for the given params")
        params.check()                                          # RawBytes: points off the curve fail the read
        if params.k > k:
            params.downsize(k)
    else:
        tau = int.from_bytes(os.urandom(64), "little") % R      # ParamsKZG::setup(k, OsRng)
        params = ParamsKZG.setup(k, ints_to_fr([tau]))
    params.precompute()          # the resident SRS's window tables (fixed-base commitments; params.py)
    pk, vk = keygen(params, circuit, vk_transcript_repr)
    return params, pk, vk


def synthesize_on_device(pk, tree, user_indices):
    """advice columns of `len(user_indices)` inclusion circuits in one launch: a (users, 3, 32 n) uint8 tensor.  The floor
    plan enters as mst_inclusion.witness_program (cached on the proving key, resident on the device)."""
    import ctypes as C
    import torch
    from . import ffi
    levels, nc, nb = pk.circuit_shape
    if (tree.depth, tree.n_currencies) != (levels, nc):
        raise ValueError("the proving key was generated for a circuit of other dimensions")
    cached = getattr(pk, "_witness_program", None)
    if cached is None:
        with getattr(pk, "_lock", P._KEY_LOCK):      # proofs in flight start together on a fresh key: built once
            cached = getattr(pk, "_witness_program", None)
            if cached is None:
                prog, n_items, n_absorbs, _, rows_used = M.witness_program(pk.k, levels, nc, nb)
                cached = pk._witness_program = (torch.from_numpy(prog.view(np.int32).copy()).cuda(), n_items, n_absorbs)
    d_prog, n_items, n_absorbs = cached
    idx = torch.tensor([int(i) for i in user_indices], dtype=torch.int32, device="cuda")
    advice = torch.empty((len(user_indices), 3, 32 * pk.n), dtype=torch.uint8, device="cuda")
    ffi.check(ffi.lib().sg_mst_inclusion_witness_dev(
        ffi.dev_ptr(d_prog), C.c_uint32(n_items), C.c_uint32(n_absorbs), ffi.dev_ptr(tree.d_users), ffi.dev_ptr(tree.d_h),
        ffi.dev_ptr(tree.d_b), C.c_uint32(levels), C.c_uint32(nc), ffi.dev_ptr(idx), C.c_uint32(len(user_indices)),
        ffi.dev_ptr(advice), C.c_uint64(pk.n), ffi.current_stream_ptr()))
    return advice


def _advice_columns(pk, circuit: MstInclusionCircuit):
    if getattr(pk, "circuit_shape", circuit.shape()) != circuit.shape():
        raise ValueError("the proving key was generated for a circuit of other dimensions")
    if circuit._prefetched is not None:
        return list(circuit._prefetched[0])
    if circuit._device is not None:
        adv = synthesize_on_device(pk, circuit._device[0], [circuit._device[1]])
        return [adv[0, j] for j in range(3)]
    asg = circuit.synthesize(pk.k)
    return [_device_column(c, pk.n) for c in asg["advice"]]


DRIVER = os.environ.get("SUMMA_PROVER_DRIVER", "native")    # "native": the library's compiled host driver; "python": prover.create_proof
_tls = threading.local()      # .combine: this thread's proofs may share fused commitment jobs with other threads' (batch.py sets it)


def set_commit_combining(on: bool) -> None:
    """for the calling thread: let its proofs' commitment jobs be fused with those of proofs in flight on other threads
    (the library's commit combiner; off by default -- a lone proof gains nothing from the bounded wait it implies)"""
    _tls.combine = bool(on)



def _create_proof(params, pk, circuit, instances, flavour: str) -> bytes:
    if len(instances) != 1:
        raise ValueError("one instance column expected")
    inst = [int(v) % R for v in instances[0]]
    advice = _advice_columns(pk, circuit)
    if DRIVER == "native":
        # device-synthesized columns are fresh per proof: the prover may write its blinding rows into them
        return P.create_proof_native(params, pk, advice, inst, flavour, sanity_checks=False, in_place=circuit._device is not None,
                                     combine=getattr(_tls, "combine", False))
    transcript = P.EvmTranscriptWriter() if flavour == "evm" else P.Blake2bWrite()
    return P.create_proof(params, pk, advice, inst, transcript=transcript, sanity_checks=False)


def full_prover(params: ParamsKZG, pk, circuit: MstInclusionCircuit, public_inputs) -> bytes:
    """[REF utils.rs:82-107]  `create_proof::<KZG, ProverSHPLONK, Challenge255, OsRng, Blake2bWrite, _>`: the proof
    under the Blake2b transcript (1632 bytes for this circuit: 16 compressed points + 35 scalars).  `public_inputs`:
    [[values of the instance column]], as `circuit.instances()` returns them.  A witness that violates a copy or a
    lookup constraint raises ("prover should not fail"); a violated gate yields a proof the verifier rejects."""
    return _create_proof(params, pk, circuit, public_inputs, "blake2b")


def full_verifier(params: ParamsKZG, vk: VerifyingKey, proof: bytes, public_inputs) -> bool:
    """[REF utils.rs:110-131]  `verify_proof::<KZG, VerifierSHPLONK, Challenge255, Blake2bRead, SingleStrategy>`"""
    from . import verifier as V
    if len(public_inputs) != 1:
        return False
    return V.verify_proof(params, vk, proof, [int(v) for v in public_inputs[0]], flavour="blake2b")


def create_proof_checked(params: ParamsKZG, pk, circuit: MstInclusionCircuit, instances) -> bytes:
    """[REF utils.rs:162-196]  the proof under the Keccak transcript of the Solidity verifier, verified right away"""
    from . import verifier as V
    proof = _create_proof(params, pk, circuit, [instances], "evm")
    assert V.verify_proof(params, pk.vk, proof, [int(v) for v in instances], flavour="evm")
    return proof


def gen_proof_solidity_calldata(params: ParamsKZG, pk, circuit: MstInclusionCircuit):
    """[REF utils.rs:134-160]  -> (proof bytes, [U256 public inputs]): what `encode_calldata` packs and the ABI decode
    of `verifyProof(bytes proof, uint256[] instances)` hands back -- the proof bytes and the canonical integers"""
    instances = circuit.instances()[0]
    proof = create_proof_checked(params, pk, circuit, instances)
    return proof, [field_element_to_solidity_calldata(v) for v in instances]


def field_element_to_solidity_calldata(field_element: int) -> int:
    """[REF utils.rs:199-203]  `U256::from_little_endian(fe.to_repr())`: the canonical integer"""
    return int(field_element) % R
