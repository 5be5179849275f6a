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


# ---------------------------------------------------------------------------------------------------------
# Levelling the ranks' particle counts (mpi_mod.f90:566-856).  The plan is the library's (fpx_redist_plan, pure host
# arithmetic); the transport is the host's -- here torch.distributed send/recv of the ONE message the engine packs.
# ---------------------------------------------------------------------------------------------------------
def redist_plan(counts, rank: int, ipout: int = 1, lib=None):
    """(role, peer, num_trans) of `rank` for the particle counts of all ranks: role 0 = nothing, 1 = send, 2 = receive."""
    import ctypes as C
    if lib is None:
        from . import _lib
        lib = _lib.load()
    counts = [int(c) for c in counts]
    arr = (C.c_int64 * len(counts))(*counts)
    role, peer, nt = C.c_int32(0), C.c_int32(-1), C.c_int64(0)
    rc = lib.fpx_redist_plan(arr, len(counts), int(rank), int(ipout), C.byref(role), C.byref(peer), C.byref(nt))
    if rc != 0:
        raise ValueError(f"fpx_redist_plan: status {rc}")
    return int(role.value), int(peer.value), int(nt.value)


def redistribute_particles(dist, eng, itime: int, ipout: int = 1):
    """mpif_calculate_part_redist + mpif_redist_part for one engine per rank: all-gather of numpart, the plan, and one
    send / recv of the packed particles between the two ranks of a pair.  Returns (role, peer, num_trans)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = torch.tensor([int(eng.n)], dtype=torch.int64)
    allc = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allc, mine)                                   # MPI_Allgather(numpart ...), mpi_mod.f90:603
    role, peer, nt = redist_plan([int(t.item()) for t in allc], rank, ipout, eng.lib)
    if role == 1:
        buf = eng.redist_pack(itime, nt)
        dist.send(torch.from_numpy(np.ascontiguousarray(buf)), dst=peer)
    elif role == 2:
        nb = int(eng.lib.fpx_redist_bytes(eng.h, nt))
        t = torch.zeros(nb, dtype=torch.uint8)
        dist.recv(t, src=peer)
        eng.redist_unpack(itime, nt, t.numpy())
    return role, peer, nt
