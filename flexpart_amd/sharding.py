"""Particle sharding across the GPUs of a node.

Particles are independent within a step, so every rank owns a contiguous range of particle
numbers `[g*N/G, (g+1)*N/G)` (the reference's MPI layout: equal split, `mpi_mod.f90:323`), the met
fields are replicated, and the only exchange is the sum of the sampling grids at output times
(`mpi_mod.f90:2471-2492`).  This module holds the host logic shared by `bench.py` and the tests;
it works with any `torch.distributed` backend (RCCL on GPUs, gloo on CPU).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous range of particle numbers owned by `rank`: [lo, hi)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return (n * rank) // world, (n * (rank + 1)) // world


def shard_scenario(sc: dict, world: int, rank: int) -> dict:
    """The scenario restricted to this rank's particles (fields are shared, not copied)."""
    n = int(sc["npart"])
    lo, hi = shard_bounds(n, world, rank)
    out = dict(sc)
    out["npart"] = hi - lo
    for k in ("xtra1", "ytra1", "ztra1", "itra1", "itramem", "npoint", "nclass", "idt", "uap", "ucp",
              "uzp", "us", "vs", "ws", "cbt"):
        if k in sc:
            out[k] = np.asarray(sc[k])[lo:hi]
    if "xmass1" in sc:
        out["xmass1"] = np.asarray(sc["xmass1"]).reshape(int(sc["nspec"]), n)[:, lo:hi]
    # the release-point tables describe the whole run (npart(kp) = particles of point kp on ALL ranks): the
    # mass-fraction termination must not depend on the number of ranks
    if "npart_rel" not in sc:
        out["npart_rel"] = np.full(int(sc.get("numpoint", 1)), max(n, 1), np.int32)
    out["particle_base"] = lo
    return out


def share_unique_id(dist, make_id, src: int = 0) -> bytes:
    """Rank `src` creates the RCCL unique id (`Engine.comm_unique_id`), everyone receives it."""
    obj = [make_id() if dist.get_rank() == src else None]
    dist.broadcast_object_list(obj, src=src)
    return obj[0]


def reduce_output_grids(dist, partial: dict) -> dict:
    """The grid reduction at an output time for a host that keeps the per-rank grids in host arrays (the protocol
    of `mpi_mod.f90:2451-2492` / `fpx_get_grids(allreduce=1)`): every array of `partial` is summed over the ranks
    into a NEW array -- the partial sums are left untouched, because the deposition grids keep accumulating and are
    reduced again at the next output time; only gridunc (and creceptor) are zeroed by the caller after writing."""
    return {k: allreduce_sum_numpy(dist, v) for k, v in partial.items()}


def allreduce_sum_numpy(dist, a: np.ndarray) -> np.ndarray:
    """Sum of a host array over all ranks (CPU path of the grid reduction, e.g. gloo)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()
