"""Proof-level batch driver: inclusion proofs for many users of one snapshot, across the GPUs of a node.

What the reference's backend does one call at a time [REF backend/src/apis/round.rs:132-174: `Snapshot::new` generates
the setup artifacts once, `generate_proof_of_inclusion(user_index)` = `gen_proof_solidity_calldata` per user], as a
batch (BASELINE configs[4]; SURVEY.md §8e level 1, "proof-level"):

* one process per GPU (`torch.distributed`; backend nccl = RCCL on the GPUs, gloo in the CPU tests);
* the setup artifacts exist once: rank 0 loads / generates the SRS and the proving key's Lagrange columns and
  broadcasts them (`broadcast_setup`: one `dist.broadcast` per buffer -- 2 * 64 * 2^k bytes of SRS, 17 * 32 * 2^k
  bytes of key columns; with nccl from device tensors into device tensors, no host copy on either side); every rank
  derives the key's coefficient / extended forms on its own device;
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

from . import api, ffi


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
_HEAD_BYTES = 64 + 128 + 128 + 32      # 8 x int64 (k, levels, nc, nb, #fixed, #sigma) | g2 | s_g2 | vk digest (little-endian)


def _as_bytes_tensor(buf, device: str):
    """a uint8 tensor on `device` holding `buf` (numpy / bytes / torch tensor); a tensor already there is used as it is"""
    import torch
    if isinstance(buf, torch.Tensor):
        t = buf.reshape(-1).view(torch.uint8) if buf.dtype != torch.uint8 else buf.reshape(-1)
        return t if t.device.type == device else t.to(device)
    return torch.from_numpy(np.array(np.frombuffer(bytes(buf), dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else buf,
                                     dtype=np.uint8, copy=True).reshape(-1)).to(device)


def broadcast_setup(setup, src: int = 0):
    """`setup`: on rank `src` a dict {k, shape (levels, n_currencies, n_bytes), g, g_lagrange (2^k x 64 B), g2, s_g2 (128 B),
    fixed [11], sigma [6] (Lagrange columns, 2^k x 32 B), vk_digest}; None elsewhere.  Returns the same dict on every rank.
    One small header broadcast, then one `dist.broadcast` per buffer (their sizes follow from k).

    With the nccl backend the buffers are DEVICE tensors on both sides and stay there: rank `src` sends straight from the
    tensors its proving key keeps (`export_setup(..., on_device=True)`), the other ranks receive into the tensors theirs
    will keep (`import_setup` hands them to the library device to device) -- the artifacts travel GPU to GPU over xGMI
    and never visit a host.  With gloo (CPU tests, rehearsals) they are host arrays."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return setup
    import torch
    on_device = d.get_backend() == "nccl"
    device = "cuda" if on_device else "cpu"
    rank = d.get_rank()
    head = torch.zeros(_HEAD_BYTES, dtype=torch.uint8, device="cpu")
    staged, failure = {}, None
    if rank == src:
        # everything that can be wrong with the source's data is found HERE, before the first collective: the header
        # carries a status word, and a source that cannot send says so in the one broadcast every rank takes part in
        # (a source that raised between two broadcasts would leave the other ranks waiting in the next one)
        try:
            levels, nc, nb = setup["shape"]
            k = int(setup["k"])
            if not 1 <= k <= 28:
                raise ValueError(f"broadcast_setup: k = {k}")
            sizes = [("g", setup["g"], 64 << k), ("g_lagrange", setup["g_lagrange"], 64 << k)]
            sizes += [(f"fixed[{j}]", c, 32 << k) for j, c in enumerate(setup["fixed"])]
            sizes += [(f"sigma[{j}]", c, 32 << k) for j, c in enumerate(setup["sigma"])]
            for name, buf, nbytes in sizes:
                t = _as_bytes_tensor(buf, device)
                if t.numel() != nbytes:
                    raise ValueError(f"broadcast_setup: {name} has {t.numel()} bytes, k = {k} needs {nbytes}")
                staged[name] = t
            ints = np.array([k, levels, nc, nb, len(setup["fixed"]), len(setup["sigma"]), 0, 0], dtype="<i8")
            raw = ints.tobytes() + bytes(setup["g2"]).ljust(128, b"\0")[:128] + bytes(setup["s_g2"]).ljust(128, b"\0")[:128] + \
                int(setup["vk_digest"]).to_bytes(32, "little")
            head = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy())
        except Exception as ex:   # noqa: BLE001 -- carried to every rank as the status word
            failure = ex
            ints = np.array([0, 0, 0, 0, 0, 0, 1, 0], dtype="<i8")                    # word 6: status (0 = ok)
            head = torch.from_numpy(np.frombuffer(ints.tobytes().ljust(_HEAD_BYTES, b"\0"), dtype=np.uint8).copy())
    head = head.to(device)
    d.broadcast(head, src)
    raw = bytes(head.cpu().numpy())
    k, levels, nc, nb, n_fixed, n_sigma, status = (int(v) for v in np.frombuffer(raw[:56], dtype="<i8"))
    if status:
        raise RuntimeError(f"broadcast_setup: rank {src} has nothing valid to send") from failure
    out = {"k": k, "shape": (levels, nc, nb), "g2": raw[64:192], "s_g2": raw[192:320],
           "vk_digest": int.from_bytes(raw[320:352], "little")}

    def one(name, nbytes):
        t = staged[name] if rank == src else torch.empty(nbytes, dtype=torch.uint8, device=device)
        d.broadcast(t, src)
        return t if on_device else t.numpy()

    for name in ("g", "g_lagrange"):
        out[name] = one(name, 64 << k)
    out["fixed"] = [one(f"fixed[{j}]", 32 << k) for j in range(n_fixed)]
    out["sigma"] = [one(f"sigma[{j}]", 32 << k) for j in range(n_sigma)]
    return out


def export_setup(params, pk, on_device: bool = False) -> dict:
    """the broadcastable form of (params, pk).  on_device: the buffers as device tensors -- the key columns are the very
    tensors `pk` keeps, the bases are device copies of the library's resident SRS -- for a broadcast that stays on the
    GPUs; otherwise host arrays"""
    if on_device:
        d_g, d_gl = params.device_bases()
        cols = lambda cs: list(cs)
    else:
        d_g, d_gl = params.g, params.g_lagrange
        cols = lambda cs: [c.cpu().numpy() for c in cs]
    return {"k": pk.k, "shape": tuple(pk.circuit_shape), "g": d_g, "g_lagrange": d_gl,
            "g2": bytes(params.g2), "s_g2": bytes(params.s_g2),
            "fixed": cols(pk.fixed_lagrange), "sigma": cols(pk.sigma_lagrange), "vk_digest": pk.vk_digest}


def import_setup(setup):
    """(params, pk, vk) on this rank's device from a broadcast setup dict; device tensors in it are used where they are"""
    import torch
    from . import prover as P
    from .params import ParamsKZG
    if isinstance(setup["g"], torch.Tensor) and setup["g"].is_cuda:
        params = ParamsKZG.from_device(setup["k"], setup["g"], setup["g_lagrange"], bytes(setup["g2"]), bytes(setup["s_g2"]))
    else:
        params = ParamsKZG(setup["k"], np.asarray(setup["g"]), np.asarray(setup["g_lagrange"]), bytes(setup["g2"]), bytes(setup["s_g2"]))
    params.precompute()
    dev = lambda a: a if (isinstance(a, torch.Tensor) and a.is_cuda) else torch.from_numpy(np.ascontiguousarray(a)).cuda()
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
    import torch
    d = _dist()
    on_device = d.get_backend() == "nccl"
    # rank 0's outcome first: if key generation fails there, every rank raises instead of waiting in a broadcast
    made, failure = None, None
    if rank == 0:
        try:
            made = api.generate_setup_artifacts(k, params_path, api.MstInclusionCircuit.init_empty(levels, n_currencies, n_bytes),
                                                vk_transcript_repr)
        except Exception as ex:
            failure = ex
    status = torch.tensor([0 if failure is None else 1], dtype=torch.int32, device="cuda" if on_device else "cpu")
    d.broadcast(status, 0)
    if int(status.item()):
        raise RuntimeError("generate_setup_artifacts failed on rank 0") from failure
    if rank == 0:
        broadcast_setup(export_setup(made[0], made[1], on_device=on_device), 0)
        return made
    return import_setup(broadcast_setup(None, 0))


# ---------------------------------------------------------------------------------------------------------------------
_WORKERS = {}            # in_flight -> ThreadPoolExecutor, kept: a worker thread's session in the library (streams, buffer pool,
_WORKERS_LOCK = threading.Lock()   # events; include/summa_prover.hpp) lives as long as the thread does, so the threads are reused


def _workers(in_flight: int) -> ThreadPoolExecutor:
    with _WORKERS_LOCK:
        pool = _WORKERS.get(in_flight)
        if pool is None:
            pool = _WORKERS[in_flight] = ThreadPoolExecutor(max_workers=in_flight, thread_name_prefix=f"prove{in_flight}",
                                                             initializer=ffi.bind_thread)   # rank r's threads on device r
        return pool


class _WitnessAhead:
    """`Circuit::synthesize` for the batch's users a chunk at a time, ahead of the provers: one launch of the witness kernel
    lays out the advice columns of `chunk` users of a device-resident snapshot (sg_mst_inclusion_witness_dev takes any
    number of users; its duration is the latency of one Poseidon sponge chain, about a millisecond, whatever the count),
    and one gather brings their public inputs.  Per proof that is 1 / chunk of a launch instead of a launch of its own.
    A background thread keeps at most `ahead` chunks laid out beyond what the provers have taken."""

    def __init__(self, tree, pk, users, chunk: int = 16, ahead: int = 3):
        import torch
        self.tree, self.pk, self.users = tree, pk, list(users)
        self.chunk, self.ahead = chunk, ahead
        self.ready, self.error = {}, None
        self.cv = threading.Condition()
        self.taken = 0
        self.closed = False
        self.stream = torch.cuda.Stream()
        self.thread = threading.Thread(target=self._run, name="witness-ahead", daemon=True)
        self.thread.start()

    def _run(self):
        import torch
        try:
            ffi.bind_thread()
            with torch.cuda.stream(self.stream):
                for first in range(0, len(self.users), self.chunk):
                    with self.cv:
                        self.cv.wait_for(lambda: self.closed or first < self.taken + self.ahead * self.chunk)
                        if self.closed:
                            return
                    part = self.users[first:first + self.chunk]
                    adv = api.synthesize_on_device(self.pk, self.tree, part)
                    inst = self.tree.public_inputs_many(part)
                    self.stream.synchronize()          # the provers read the columns on their own streams
                    with self.cv:
                        for j, u in enumerate(part):
                            self.ready[u] = ([adv[j, c] for c in range(3)], inst[j])
                        self.cv.notify_all()
        except Exception as ex:   # handed to whoever waits
            with self.cv:
                self.error = ex
                self.cv.notify_all()

    def close(self):
        """the batch is over (or failed): the producer stops, what it had laid out is dropped"""
        with self.cv:
            self.closed = True
            self.ready.clear()
            self.cv.notify_all()

    def circuit(self, user: int):
        with self.cv:
            self.cv.wait_for(lambda: user in self.ready or self.error is not None or self.closed)
            if user not in self.ready:
                self.taken += 1          # the producer's look-ahead window moves on past a user nobody will get
                self.cv.notify_all()
                raise self.error if self.error is not None else RuntimeError("witness producer closed (the batch was abandoned)")
            got = self.ready.pop(user)
            self.taken += 1
            self.cv.notify_all()
        c = api.MstInclusionCircuit.init_from_tree(self.tree, user)
        c._prefetched = got
        return c


class BatchResult:
    def __init__(self):
        self.proofs = {}          # user index -> (proof bytes, public inputs)
        self.seconds = 0.0
        self.errors = {}
        self.wait_sleep_us = None  # how this batch's host threads waited (host.wait_sleep_us while it ran; None: not a GPU batch)

    def proofs_per_s(self) -> float:
        return len(self.proofs) / self.seconds if self.seconds else 0.0


def prove_batch(tree, user_indices, params, pk, levels: int, flavour: str = "evm", in_flight: int = 2, prove=None,
                make_circuit=None, combine: bool | None = None) -> BatchResult:
    """Inclusion proofs for this rank's share of `user_indices` (dealt round-robin over the process group).

    tree: MerkleSumTree of the snapshot (every rank holds it -- it is the input data); flavour "evm" =
    `gen_proof_solidity_calldata` per user (what the backend serves), "blake2b" = `full_prover`.
    in_flight: proofs in flight on this GPU, each on its own stream / worker thread.
    `prove(circuit) -> (proof, public_inputs)` and `make_circuit(user_index)` replace the default steps (the CPU tests
    of the scheduling inject stand-ins; the product path uses the API functions).
    combine (default: in_flight >= 6; SUMMA_COMBINE_COMMITS=0/1 overrides): the commitment jobs of the proofs in flight are
    fused by the library's commit combiner (include/summa_gpu.h: sg_commit_combine_begin) -- one sort front-end, bucket
    reduction and host tail per job for all of them.  It pays once enough proofs are in flight for full jobs to form
    while one is running (profiles/r03_sweeps/commit_combiner.txt: 4 in flight 184 -> 174 proofs/s, 8: 189 -> 217,
    16: 179 -> 227)."""
    import os
    if combine is None:
        env = os.environ.get("SUMMA_COMBINE_COMMITS")
        combine = (env == "1" or (env is None and in_flight >= 6)) and in_flight > 1 and prove is None
    # process-wide tunables of the library for the length of this batch.  What was found is put back when the LAST batch of the
    # process ends (_ParamScope counts them): a caller's own settings -- SG_PARAMS, bench.py's multi-rank wait -- survive, and a
    # batch that finishes does not switch another one, still running, back to busy polling
    scope = {}
    if combine:
        # SUMMA_COMBINE_TARGET / SUMMA_COMBINE_WAIT_US override them for sweeps (at 64 in flight: targets 16 / 32 / 64 and waits
        # of 1 / 2 / 5 ms all land within the run-to-run spread, 249-266 proofs/s)
        scope["commit.combine_target"] = int(os.environ.get("SUMMA_COMBINE_TARGET", in_flight))
        scope["commit.combine_wait_us"] = int(os.environ.get("SUMMA_COMBINE_WAIT_US", 5000))
        # two fused jobs side by side (round 5): one job's sort front end, bucket reduction and host tail run under the other's
        # accumulation -- with one runner an accumulation is on the device half of the time only (profiles/r04_sweeps/
        # batch_concurrency_64_end_of_round.txt).  Measured at 64 in flight: 249-256 -> 272-274 proofs/s on a whole host, 272 on a 1/8
        # share of it (profiles/r05_sweeps/batch_knobs.txt); with few proofs in flight jobs of two runners are too small to pay
        if in_flight >= 16 or "SUMMA_COMBINE_RUNNERS" in os.environ:
            scope["commit.combine_runners"] = int(os.environ.get("SUMMA_COMBINE_RUNNERS", 2))
    # with several proofs in flight most worker threads are waiting for the device most of the time: they poll and SLEEP
    # (sg_set_param "host.wait_sleep_us") instead of polling and yielding -- the same proofs per second on a whole host, a
    # fifth more on a 1/8 share of it (what a rank gets when eight share a node), 13 -> 8 ms of CPU per proof
    # (profiles/r04_sweeps/batch_wait_modes.txt).  A lone proof keeps the runtime's wait: its dozen waits are on its critical path.
    # SUMMA_WAIT_SLEEP_US=N overrides (0: keep polling); a wait the caller has already made sleep (SG_PARAMS, sg_set_param) is kept
    nap_env = os.environ.get("SUMMA_WAIT_SLEEP_US")
    if prove is None and in_flight >= 4 and nap_env != "0":
        if nap_env is not None:
            scope["host.wait_sleep_us"] = int(nap_env)
        elif ffi.get_param("host.wait_sleep_us") == 0:
            scope["host.wait_sleep_us"] = 50
    params_scope = _ParamScope(scope)
    params_scope.enter()
    mine = deal(list(user_indices))
    res = BatchResult()
    if prove is None:
        res.wait_sleep_us = ffi.get_param("host.wait_sleep_us")
    ahead = None
    if make_circuit is None:
        if hasattr(tree, "d_h"):      # a device-resident snapshot: the witness never visits the host
            if len(mine) >= 8 and len(set(mine)) == len(mine) and hasattr(pk, "circuit_shape"):
                ahead = _WitnessAhead(tree, pk, mine)       # chunks of users per witness launch, ahead of the provers
                make_circuit = ahead.circuit
            else:
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
            api.set_commit_combining(combine)
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
    try:
        if in_flight <= 1:
            for i in mine:
                work(i)
        else:
            list(_workers(in_flight).map(work, mine))
    finally:
        if ahead is not None:
            ahead.close()
        params_scope.leave()     # a later caller's lone proofs do not wait 5 ms for company, nor sleep between polls
    res.seconds = time.perf_counter() - t0
    return res


class _ParamScope:
    """Library parameters held at given values while at least one batch of this process runs.  The first batch to enter saves
    what it finds (sg_get_param), every batch sets its own values, the last one to leave restores the saved ones."""
    _lock = threading.Lock()
    _active = 0
    _saved: dict = {}

    def __init__(self, values):
        self.values = dict(values)

    def enter(self):
        cls = _ParamScope
        with cls._lock:
            for name, value in self.values.items():
                if name not in cls._saved:
                    cls._saved[name] = ffi.get_param(name)
                ffi.set_param(name, value)
            cls._active += 1

    def leave(self):
        cls = _ParamScope
        with cls._lock:
            cls._active -= 1
            if cls._active == 0:
                for name, value in cls._saved.items():
                    ffi.set_param(name, value)
                cls._saved = {}


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
