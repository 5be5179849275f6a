"""CPU: the N>1 host path (particle sharding + grid reduction) with world_size-2 gloo."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from flexpart_amd import sharding
from flexpart_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition_everything():
    for n in (0, 1, 7, 1000, 10**8):
        for world in (1, 2, 3, 8):
            edges = [sharding.shard_bounds(n, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 2, 2)


def test_shard_scenario_slices_particles_only():
    sc = syn.small(n=101, nx=20, ny=12, nz=10)
    a = sharding.shard_scenario(sc, 2, 0)
    b = sharding.shard_scenario(sc, 2, 1)
    assert a["npart"] + b["npart"] == 101
    assert np.array_equal(np.concatenate([a["xtra1"], b["xtra1"]]), sc["xtra1"])
    assert a["uu"] is sc["uu"]      # fields are replicated by reference, not copied


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch.distributed as dist
    from flexpart_amd import sharding, synthetic as syn
    from oracle.oracle import Oracle
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
    rank = dist.get_rank()
    sc = syn.add_outgrid(syn.small(n=600, nx=30, ny=20, nz=16, nsteps=1, ctl=-5.0, hmix_const=100.0,
                                   frac_pbl=0.0, turb_off=True), nxg=12, nyg=8, nzg=3)
    # every rank samples ITS shard into a private grid; the grids are then summed (the collective
    # of mpi_mod.f90:2471-2492); the id hand-off used for RCCL is exercised with a dummy id
    mine = sharding.shard_scenario(sc, 2, rank)
    o = Oracle(mine, "r8")
    o.sample()            # conccalc over this rank's particles (no RNG involved: shard == serial)
    g, _ = o.grids()
    total = sharding.allreduce_sum_numpy(dist, g)
    uid = sharding.share_unique_id(dist, lambda: b"x" * 128)
    assert uid == b"x" * 128
    if rank == 0:
        full = Oracle(sc, "r8")
        full.sample()
        fg, _ = full.grids()
        assert np.abs(total - fg).max() <= 1e-12 * fg.max(), np.abs(total - fg).max()
        assert abs(total.sum() - fg.sum()) <= 1e-9 * fg.sum()
        print("OK", total.sum())
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_grid_reduction(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0]
