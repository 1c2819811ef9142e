"""Multi-GPU sharding of the path: one process per GPU, torch.distributed.

* proof-level: whole proofs of a batch are dealt round-robin to ranks -- no data-path
  communication (`batch.deal`).  The 16 MSMs + 19 NTTs of ONE proof are not spread over GPUs:
  DESIGN.md section 5 counts what that would move (under 0.5 ms of a 5.7 ms proof on two GPUs,
  against a second whole proof on the second GPU).
* point-level: one large MSM is cut into contiguous shards; every rank reduces its shard and
  the 64-byte affine partials are exchanged with ONE all_gather (EC addition is not an RCCL
  reduction operator, so the "all-reduce" is all_gather + a local sum of world_size points,
  done as an n = world_size MSM with unit scalars).
"""
from __future__ import annotations

import numpy as np

from .arithmetic import best_multiexp

# Montgomery form of 1 in Fr (R mod r), 32 bytes little-endian
_ONE_FR = np.frombuffer((0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb).to_bytes(32, "little"),
                        dtype=np.uint8)


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def shard_bounds(n: int, rank: int, world: int):
    """contiguous shard [lo, hi) of an n-point MSM for `rank`"""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def combine_partials(partials: np.ndarray, msm=None) -> np.ndarray:
    """sum of world_size affine points (64 B each).  Product path: a few host-side point
    additions in the library (sg_g1_sum_affine).  `msm`: injected MSM callable (CPU tests run
    the collective logic with the oracle standing in for the GPU): sum = MSM with unit scalars."""
    m = partials.size // 64
    if msm is not None:
        return msm(np.tile(_ONE_FR, m), partials)
    import ctypes as C
    from . import ffi
    out = np.zeros(64, dtype=np.uint8)
    p = np.ascontiguousarray(partials)
    ffi.check(ffi.lib().sg_g1_sum_affine(ffi.ptr(p), C.c_size_t(m), ffi.ptr(out)))
    return out


def exchange_partials(part: np.ndarray, msm=None) -> np.ndarray:
    """the exchange step of a point-sharded MSM: every rank contributes its 64-byte partial point, ONE all_gather
    (RCCL over xGMI with the nccl backend), a local sum of world_size points; every rank returns the same point.
    With no process group: the partial itself."""
    d = _dist()
    if d is None:
        return part
    import torch        # with a process group the exchange step always runs (world 1 included: same code path at every N)
    world = d.get_world_size()
    dev = "cuda" if d.get_backend() == "nccl" else "cpu"
    mine = torch.from_numpy(np.ascontiguousarray(part)).to(dev)
    gathered = torch.empty(64 * world, dtype=torch.uint8, device=dev)
    d.all_gather_into_tensor(gathered, mine)
    return combine_partials(gathered.cpu().numpy(), msm)


def exchange_partials_many(parts, msm=None):
    """the exchange step of SEVERAL point-sharded MSMs in ONE collective: every rank contributes the 64-byte partials of
    its K steps (they are host-resident: the MSM's tail runs on the host), one all_gather of K * 64 bytes, K local sums
    of world_size points.  Returns the K result points (the same on every rank).

    Why a pipelined caller uses this instead of K calls of `exchange_partials`: with steps in flight, accumulations of
    consecutive MSMs run back to back on the device, and every kernel beside them must raise its wave priority to be
    issued at all (csrc/side_prio.cuh); RCCL's kernels cannot.  One collective after the steps
    keeps RCCL off the device while it is saturated; the step's result is complete when this returns."""
    parts = [np.ascontiguousarray(p_) for p_ in parts]
    d = _dist()
    if d is None or not parts:
        return parts
    import torch
    world = d.get_world_size()
    dev = "cuda" if d.get_backend() == "nccl" else "cpu"
    K = len(parts)
    mine = torch.from_numpy(np.concatenate(parts)).to(dev)
    gathered = torch.empty(64 * K * world, dtype=torch.uint8, device=dev)
    d.all_gather_into_tensor(gathered, mine)
    g = gathered.cpu().numpy().reshape(world, K, 64)
    return [combine_partials(np.ascontiguousarray(g[:, i, :]).reshape(-1), msm) for i in range(K)]


def sharded_msm(local_scalars, local_bases, msm=None) -> np.ndarray:
    """sum over ALL ranks' shards of sum_i s_i P_i; every rank returns the same 64-byte point.
    With no process group this is plain best_multiexp."""
    return exchange_partials((msm or best_multiexp)(local_scalars, local_bases), msm)
