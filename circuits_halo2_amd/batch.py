"""Proof-level batch driver: inclusion proofs for many users of one snapshot, across the GPUs of a node.

What the reference's backend does one call at a time [REF backend/src/apis/round.rs:132-174: `Snapshot::new` generates
the setup artifacts once, `generate_proof_of_inclusion(user_index)` = `gen_proof_solidity_calldata` per user], as a
batch (BASELINE configs[4]; SURVEY.md §8e level 1, "proof-level"):

* one process per GPU (`torch.distributed`; backend nccl = RCCL on the GPUs, gloo in the CPU tests);
* the setup artifacts exist once: rank 0 loads / generates the SRS and the proving key's Lagrange columns and
  broadcasts them (`broadcast_setup`: one `dist.broadcast` per buffer -- 2 * 64 * 2^k bytes of SRS, 17 * 32 * 2^k
  bytes of key columns); every rank derives the key's coefficient / extended forms on its own device;
* users are dealt round-robin to ranks (`deal`): proofs are independent, so there is no data-path collective at
  all -- the only other communication is the optional gather of the finished proofs (`gather_proofs`, host bytes);
* per GPU several proofs are in flight: worker threads, each on its own HIP stream, so that one proof's host work
  (transcript, Fiat-Shamir round trips, the MSM tails) overlaps the kernels of the others.
"""
from __future__ import annotations

import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import api


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def rank_world():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def deal(items, rank: int | None = None, world: int | None = None):
    """round-robin share of `items` for this rank: items[rank], items[rank + world], ..."""
    r, w = rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    return list(items[rank::world])


def owner_of(position: int, world: int) -> int:
    """rank that `deal` gives the item at `position` to"""
    return position % world


# ---------------------------------------------------------------------------------------------------------------------
def _bcast_bytes(buf: np.ndarray | None, src: int, device: str) -> np.ndarray:
    """broadcast a uint8 buffer whose length the other ranks do not know yet"""
    import torch
    d = _dist()
    size = torch.tensor([0 if buf is None else int(buf.size)], dtype=torch.int64, device=device)
    d.broadcast(size, src)
    t = torch.from_numpy(np.array(buf, dtype=np.uint8, copy=True)).to(device) if buf is not None else \
        torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    if int(size.item()):
        d.broadcast(t, src)
    return t.cpu().numpy()


def broadcast_setup(setup, src: int = 0):
    """`setup`: on rank `src` a dict {k, shape (levels, n_currencies, n_bytes), g, g_lagrange, g2, s_g2 (bytes),
    fixed [11], sigma [6] (Lagrange columns as uint8 buffers), vk_digest}; None elsewhere.  Returns the same dict on
    every rank.  One broadcast per buffer; with the nccl backend the buffers travel GPU to GPU over xGMI."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return setup
    import torch
    device = "cuda" if d.get_backend() == "nccl" else "cpu"
    rank = d.get_rank()
    head = torch.zeros(8, dtype=torch.int64, device=device)
    if rank == src:
        levels, nc, nb = setup["shape"]
        head[:6] = torch.tensor([setup["k"], levels, nc, nb, len(setup["fixed"]), len(setup["sigma"])], dtype=torch.int64)
    d.broadcast(head, src)
    k, levels, nc, nb, n_fixed, n_sigma = (int(v) for v in head[:6].tolist())
    mine = setup if rank == src else None
    pick = lambda name: np.frombuffer(bytes(mine[name]), dtype=np.uint8) if mine is not None else None
    out = {"k": k, "shape": (levels, nc, nb)}
    for name in ("g", "g_lagrange", "g2", "s_g2"):
        out[name] = _bcast_bytes(pick(name), src, device)
    out["fixed"] = [_bcast_bytes(np.asarray(mine["fixed"][j], dtype=np.uint8) if mine is not None else None, src, device)
                    for j in range(n_fixed)]
    out["sigma"] = [_bcast_bytes(np.asarray(mine["sigma"][j], dtype=np.uint8) if mine is not None else None, src, device)
                    for j in range(n_sigma)]
    digest = _bcast_bytes(np.frombuffer(int(mine["vk_digest"]).to_bytes(32, "little"), dtype=np.uint8) if mine is not None else None,
                          src, device)
    out["vk_digest"] = int.from_bytes(bytes(digest), "little")
    return out


def export_setup(params, pk) -> dict:
    """the broadcastable form of (params, pk): host buffers"""
    host = lambda t: t.cpu().numpy()
    return {"k": pk.k, "shape": tuple(pk.circuit_shape), "g": params.g, "g_lagrange": params.g_lagrange,
            "g2": np.frombuffer(bytes(params.g2), dtype=np.uint8), "s_g2": np.frombuffer(bytes(params.s_g2), dtype=np.uint8),
            "fixed": [host(c) for c in pk.fixed_lagrange], "sigma": [host(c) for c in pk.sigma_lagrange],
            "vk_digest": pk.vk_digest}


def import_setup(setup):
    """(params, pk, vk) on this rank's device from a broadcast setup dict"""
    import torch
    from . import prover as P
    from .params import ParamsKZG
    params = ParamsKZG(setup["k"], setup["g"], setup["g_lagrange"], bytes(setup["g2"]), bytes(setup["s_g2"]))
    params.precompute()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    levels, nc, nb = setup["shape"]
    pk = P.ProvingKey(params, setup["k"], [dev(c) for c in setup["fixed"]], [dev(c) for c in setup["sigma"]], nc)
    pk.circuit_shape = (levels, nc, nb)
    pk.vk_digest = setup["vk_digest"]
    pk.vk = api.VerifyingKey(setup["k"], nc, pk.fixed_comms, pk.permutation_comms, pk.vk_digest)
    return params, pk, pk.vk


def setup_on_all_ranks(k: int, params_path, levels: int, n_currencies: int = 2, n_bytes: int = 8, vk_transcript_repr=None):
    """`generate_setup_artifacts` once (rank 0) + broadcast: every rank returns its own (params, pk, vk)"""
    rank, world = rank_world()
    if world == 1:
        return api.generate_setup_artifacts(k, params_path, api.MstInclusionCircuit.init_empty(levels, n_currencies, n_bytes),
                                            vk_transcript_repr)
    if rank == 0:
        made = api.generate_setup_artifacts(k, params_path, api.MstInclusionCircuit.init_empty(levels, n_currencies, n_bytes),
                                            vk_transcript_repr)
        broadcast_setup(export_setup(made[0], made[1]), 0)
        return made
    return import_setup(broadcast_setup(None, 0))


# ---------------------------------------------------------------------------------------------------------------------
_WORKERS = {}            # in_flight -> ThreadPoolExecutor, kept: a worker thread's session in the library (streams, buffer pool,
_WORKERS_LOCK = threading.Lock()   # events; include/summa_prover.hpp) lives as long as the thread does, so the threads are reused


def _workers(in_flight: int) -> ThreadPoolExecutor:
    with _WORKERS_LOCK:
        pool = _WORKERS.get(in_flight)
        if pool is None:
            pool = _WORKERS[in_flight] = ThreadPoolExecutor(max_workers=in_flight, thread_name_prefix=f"prove{in_flight}")
        return pool


class BatchResult:
    def __init__(self):
        self.proofs = {}          # user index -> (proof bytes, public inputs)
        self.seconds = 0.0
        self.errors = {}

    def proofs_per_s(self) -> float:
        return len(self.proofs) / self.seconds if self.seconds else 0.0


def prove_batch(tree, user_indices, params, pk, levels: int, flavour: str = "evm", in_flight: int = 2, prove=None,
                make_circuit=None) -> BatchResult:
    """Inclusion proofs for this rank's share of `user_indices` (dealt round-robin over the process group).

    tree: MerkleSumTree of the snapshot (every rank holds it -- it is the input data); flavour "evm" =
    `gen_proof_solidity_calldata` per user (what the backend serves), "blake2b" = `full_prover`.
    in_flight: proofs in flight on this GPU, each on its own stream / worker thread.
    `prove(circuit) -> (proof, public_inputs)` and `make_circuit(user_index)` replace the default steps (the CPU tests
    of the scheduling inject stand-ins; the product path uses the API functions)."""
    mine = deal(list(user_indices))
    res = BatchResult()
    if make_circuit is None:
        if hasattr(tree, "d_h"):      # a device-resident snapshot: the witness never visits the host
            make_circuit = lambda i: api.MstInclusionCircuit.init_from_tree(tree, i)
        else:
            make_circuit = lambda i: api.MstInclusionCircuit.init(tree.generate_proof(i), levels)
    if prove is None:
        if flavour == "evm":
            prove = lambda c: api.gen_proof_solidity_calldata(params, pk, c)
        else:
            def prove(c):
                inst = c.instances()
                return api.full_prover(params, pk, c, inst), inst[0]
    lock = threading.Lock()
    local = threading.local()

    def work(i):
        try:
            import torch
            on_gpu = torch.cuda.is_available()
        except Exception:  # pragma: no cover
            on_gpu = False
        try:
            if on_gpu:
                if not hasattr(local, "stream"):
                    local.stream = torch.cuda.Stream()
                with torch.cuda.stream(local.stream):
                    out = prove(make_circuit(i))
                    local.stream.synchronize()
            else:
                out = prove(make_circuit(i))
            with lock:
                res.proofs[i] = out
        except Exception as ex:   # one bad witness must not lose the batch
            with lock:
                res.errors[i] = repr(ex)

    t0 = time.perf_counter()
    if in_flight <= 1:
        for i in mine:
            work(i)
    else:
        list(_workers(in_flight).map(work, mine))
    res.seconds = time.perf_counter() - t0
    return res


def gather_proofs(res: BatchResult, dst: int = 0):
    """all ranks' proofs on rank `dst` as {user index: (proof, public_inputs)} (host bytes; `gather_object`)"""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return dict(res.proofs)
    parts = [None] * d.get_world_size() if d.get_rank() == dst else None
    d.gather_object(res.proofs, parts, dst=dst)
    if d.get_rank() != dst:
        return None
    merged = {}
    for p in parts:
        merged.update(p)
    return merged
