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


WORKER2 = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch.distributed as dist
    from flexpart_amd import sharding, synthetic as syn
    from oracle.oracle import Oracle
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
    rank = dist.get_rank()
    # all particles above the mixing layer, turbulence parameters zero, released before the start (no initialize(),
    # whose mesoscale draw initialize.f90:207-209 is not scaled by turbmesoscale): trajectories do not depend on the
    # shared random stream, so a shard computes exactly what the serial run computes for its particles
    sc = syn.small(n=800, nx=30, ny=20, nz=16, nsteps=4, ctl=-5.0, hmix_const=100.0, frac_pbl=0.0, turb_off=True)
    sc.update(decay=np.array([1.0e-6]), xmass=np.array([1.0]), itime0=900, itra1=np.full(800, 900, np.int32),
              itramem=np.zeros(800, np.int32))
    syn.add_wet(syn.add_outgrid(sc, nxg=12, nyg=8, nzg=3), gas=False)

    def run(scn, reduce):
        o = Oracle(scn, "r8")
        outs = []
        for out_time in range(2):           # two output times, two steps each
            o.step(); o.step()
            g, d = o.grids()
            part = {"gridunc": g, "wetgridunc": o.wetgrid()}
            outs.append(reduce(part))
            o.clear_gridunc()               # concoutput.f90:719-720: gridunc only; wetgridunc accumulates on
        return outs
    mine = run(sharding.shard_scenario(sc, 2, rank), lambda p: sharding.reduce_output_grids(dist, p))
    # what an in-place reduction does (the defect of round 1): after the first output every rank's accumulator holds
    # the TOTAL; it goes on accumulating the rank's own deposits and is summed over the ranks again
    o = Oracle(sharding.shard_scenario(sc, 2, rank), "r8")
    o.step(); o.step()
    w1 = o.wetgrid()
    acc = sharding.allreduce_sum_numpy(dist, w1)
    o.step(); o.step()
    acc = acc + (o.wetgrid() - w1)
    w_inplace = sharding.allreduce_sum_numpy(dist, acc)
    if rank == 0:
        full = run(sc, lambda p: p)
        for a, b in zip(mine, full):
            for k in ("gridunc", "wetgridunc"):
                assert b[k].sum() > 0
                assert np.abs(a[k] - b[k]).max() <= 1e-6 * b[k].max(), (k, np.abs(a[k] - b[k]).max() / b[k].max())
        assert full[1]["wetgridunc"].sum() > 1.2 * full[0]["wetgridunc"].sum()      # cumulative over the run
        assert full[1]["gridunc"].sum() < 1.5 * full[0]["gridunc"].sum()            # per output interval, not cumulative
        # the double count this test guards against: the first total counted once per rank
        assert w_inplace.sum() > full[1]["wetgridunc"].sum() + 0.9 * full[0]["wetgridunc"].sum()
        print("OK")
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_two_output_times_keep_partial_sums(tmp_path):
    """mpi_mod.f90:2451-2492: the grids are reduced into receive arrays at EVERY output time; the deposition grids
    are cumulative over the run (zeroed only in outgrid_init.f90:317-318), gridunc is zeroed after each output
    (concoutput.f90:719-720).  Two ranks, two output times, wet deposition: the sums at both output times equal the
    serial run's.  (The engine's own implementation of this protocol -- receive buffers inside fpx_get_grids -- is
    exercised by tests/test_multirank_gpu.py with two processes on one GPU.)"""
    port = 31500 + (os.getpid() % 2000)
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2 % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0]


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("fpx_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_relaunches_itself_with_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks through torch.distributed.run -- as a child
    process, before torch is imported -- and hands its own arguments on."""
    bench = _load_bench()
    argv = bench.relaunch_argv(4, ["--gpus", "4", "--steps", "3", "--config", "4"], 29517)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=4" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29517"
    assert argv[-7] == os.path.join(ROOT, "bench.py") and argv[-6:] == ["--gpus", "4", "--steps", "3", "--config", "4"]

    calls = []

    class Done:
        returncode = 0

    def fake_run(cmd, **kw):
        calls.append(cmd)
        return Done()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--particles", "2e7"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(calls) == 1
    assert "--nproc-per-node=2" in calls[0] and calls[0][-4:] == ["--gpus", "2", "--particles", "2e7"]
    assert "torch" not in bench.__dict__        # nothing touched the GPU runtime in the launcher process


def test_bench_refuses_a_rank_count_that_differs_from_gpus(monkeypatch):
    bench = _load_bench()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "--gpus 4" in str(e.value.code) and "2 rank" in str(e.value.code)
    monkeypatch.setattr(sys, "argv", ["bench.py"])             # torchrun with two ranks but the default --gpus 1
    with pytest.raises(SystemExit):
        bench.main()
