"""GPU: the engine's own multi-rank protocol with two processes on ONE GPU.

RCCL refuses two ranks on the same device, so on a one-GPU box the reduction runs through the engine's second
transport, the host-supplied all-reduce of fpx_comm_init_host (what an MPI host passes; here gloo).  Everything else
is the code the 8-GPU run executes: particle shards with global particle numbers as RNG keys, per-rank partial sums in
device memory, receive buffers for the reduced grids, two output times with cumulative deposition grids."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COMMON = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    from flexpart_amd import sharding, synthetic as syn
    from flexpart_amd.engine import Engine, RNG_PHILOX

    def scenario():
        sc = syn.small(n=6000, nx=48, ny=32, nz=36, nsteps=4, ctl=5.0, ifine=4, cblflag=1)
        sc.update(lsettling=1, drydep=1, drydepspec=np.array([1], np.int32), density=np.array([2000.0]),
                  dquer=np.array([8.0]), vsetaver=np.array([-0.004]), cunningham=np.array([1.02]),
                  decay=np.array([1.0e-6]), xmass=np.array([1.0]))
        syn.add_outgrid(sc)
        syn.add_wet(sc, gas=False)
        syn.add_outgrid_nest(sc)
        syn.add_receptors(sc)
        return sc

    def run(eng, allreduce, sort_at=None):
        outs = []
        for out_time in range(2):               # two output times, two synchronisation steps each
            for k in range(2):
                if eng.itime != 0:
                    eng.wetdepo()
                eng.step()
                eng.conccalc(eng.itime, 1.0)
            if sort_at == out_time:
                eng.sort()
            g, d = eng.grids(allreduce=allreduce, clear=True)     # clear: gridunc only (concoutput.f90:719-720)
            w = eng.wetgrid(allreduce=allreduce)
            gn, dn, wn = eng.grids_nest(allreduce=allreduce, clear=True)
            r = eng.receptors(allreduce=allreduce, clear=True)
            outs.append(dict(gridunc=g, drygridunc=d, wetgridunc=w, griduncn=gn, drygriduncn=dn, wetgriduncn=wn, creceptor=r))
        return outs
""")

WORKER = COMMON + textwrap.dedent("""
    import torch.distributed as dist
    rank = int(sys.argv[1])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=2)
    sc = scenario()
    mine = sharding.shard_scenario(sc, 2, rank)
    eng = Engine(mine, compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_PHILOX, seed=4711, **%(ekw)r)   # particle_base from the shard
    eng.comm_init_host(dist, 2, rank)
    outs = run(eng, True, sort_at=0 if rank == 1 else None)
    state = eng.download()
    np.savez(%(out)r + f"_rank{rank}.npz", x=state["xtra1"], z=state["ztra1"], itra1=state["itra1"], m=state["xmass1"],
             blended=eng.info("blended_steps"), **{f"o{i}_{k}": v for i, o in enumerate(outs) for k, v in o.items()})
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
""")

SERIAL = COMMON + textwrap.dedent("""
    sc = scenario()
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_PHILOX, seed=4711, **%(ekw)r)
    outs = run(eng, False)
    state = eng.download()
    np.savez(%(out)r + "_serial.npz", x=state["xtra1"], z=state["ztra1"], itra1=state["itra1"], m=state["xmass1"],
             blended=eng.info("blended_steps"), **{f"o{i}_{k}": v for i, o in enumerate(outs) for k, v in o.items()})
    eng.close()
""")


@pytest.mark.parametrize("blend", ["off", "forced_on", "auto_from_the_global_count"])
def test_two_ranks_one_gpu_two_output_times(built, tmp_path, blend):
    """blend: the time-blended wind packs round differently from the plain gather, so whether a step uses them must not
    depend on the rank count: fpx_config.blend_mode (1 = on) or, in the automatic mode, the run's particle count over ALL
    ranks (global_particles: 4e7 here, above the threshold on every rank although each holds 3000 particles) -- the sharded
    and the single-rank run stay bitwise equal with the blend ON (README_PARALLEL.md:189-192), and it did run (blended_steps).
    mpi_mod.f90:2451-2492 through the engine: two ranks share one cloud (contiguous ranges of particle numbers,
    README_PARALLEL.md:60-67), reduce gridunc / drygridunc / wetgridunc (+ nested grids, creceptor) at two output
    times into receive buffers and keep accumulating their partial sums in between.  (i) With the counter RNG keyed on
    the global particle number every particle ends exactly where the single-rank run puts it.  (ii) The sums at BOTH
    output times equal the single-rank grids: an in-place reduction (round 1) fails the second one by the first total."""
    out = str(tmp_path / "mr")
    port = 33500 + (os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ekw = {"off": {}, "forced_on": {"blend_mode": 1}, "auto_from_the_global_count": {"global_particles": 40000000}}[blend]
    ws = tmp_path / "worker.py"
    ws.write_text(WORKER % dict(root=ROOT, port=port, out=out, ekw=ekw))
    ss = tmp_path / "serial.py"
    ss.write_text(SERIAL % dict(root=ROOT, out=out, ekw=ekw))
    procs = [subprocess.Popen([sys.executable, str(ws), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    r = subprocess.run([sys.executable, str(ss)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    a, b, s = (np.load(out + f"_{t}.npz") for t in ("rank0", "rank1", "serial"))
    for d in (a, b, s):
        assert (int(d["blended"]) > 0) == (blend != "off"), blend
    # (i) sharding does not change any particle (counter RNG keyed on the global number; rank 1 even re-sorted its slots)
    for k in ("x", "z", "itra1"):
        assert np.array_equal(np.concatenate([a[k], b[k]]), s[k]), k
    assert np.array_equal(np.concatenate([a["m"], b["m"]], axis=1), s["m"])
    # (ii) both output times; every rank received the same sums
    for i in range(2):
        for k in ("gridunc", "drygridunc", "wetgridunc", "griduncn", "drygriduncn", "wetgriduncn", "creceptor"):
            want, g0, g1 = s[f"o{i}_{k}"], a[f"o{i}_{k}"], b[f"o{i}_{k}"]
            assert want.sum() > 0, (i, k)
            assert np.array_equal(g0, g1), (i, k)
            tol = 1e-12 if k in ("gridunc", "griduncn", "creceptor") else 2e-5     # f32 atomics sum in another order
            assert np.abs(g0 - want).max() <= tol * want.max(), (i, k, np.abs(g0 - want).max() / want.max())
    for k in ("drygridunc", "wetgridunc", "drygriduncn", "wetgriduncn"):
        assert s[f"o1_{k}"].sum() > 1.2 * s[f"o0_{k}"].sum(), k        # the deposition grids did accumulate over the run
    assert s["o1_gridunc"].sum() < 1.5 * s["o0_gridunc"].sum()          # gridunc was zeroed after the first output


REDIST_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch.distributed as dist
    from flexpart_amd import sharding, synthetic as syn
    from flexpart_amd.engine import Engine, RNG_PHILOX
    rank = int(sys.argv[1])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=2)
    n = (260000, 120000)[rank]
    sc = syn.small(n=n, nx=48, ny=32, nz=36, nsteps=2, ctl=5.0, ifine=4, cblflag=1, seed=100 + rank)
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_PHILOX, seed=4711, max_particles=300000,
                 particle_base=rank * 300000)
    eng.upload_particles_from_scenario(sc)
    eng.step()                                   # a step first: some particles leave, the slots get sorted on rank 0
    if rank == 0:
        eng.sort()
    before = eng.download()
    itime = eng.itime
    role, peer, nt = sharding.redistribute_particles(dist, eng, itime)
    after = eng.download(0, 300000)
    stats = eng.step()                           # the received particles advance with everything else
    np.savez(%(out)r + f"_rank{rank}.npz", role=role, peer=peer, nt=nt, n_after=eng.n, itime=itime, due=stats["n_due"],
             bx=before["xtra1"], by=before["ytra1"], bz=before["ztra1"], bi=before["itra1"], bm=before["xmass1"],
             ax=after["xtra1"], ay=after["ytra1"], az=after["ztra1"], ai=after["itra1"], am=after["xmass1"])
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_ranks_level_their_particle_counts(built, tmp_path):
    """mpif_calculate_part_redist + mpif_redist_part (mpi_mod.f90:566-856) between two processes on one GPU, the transport
    being the host's (gloo send / recv of the one packed message): 260000 and 120000 particles -> the plan moves 70000; the
    particles alive at itime are the same set before and after (position, height and mass of each), the counts are levelled,
    and the next step advances every one of them."""
    out = str(tmp_path / "rd")
    port = 35500 + (os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ws = tmp_path / "redist_worker.py"
    ws.write_text(REDIST_WORKER % dict(root=ROOT, port=port, out=out))
    procs = [subprocess.Popen([sys.executable, str(ws), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    a, b = (np.load(out + f"_rank{r}.npz") for r in (0, 1))
    assert (int(a["role"]), int(a["peer"]), int(a["nt"])) == (1, 1, 70000)
    assert (int(b["role"]), int(b["peer"]), int(b["nt"])) == (2, 0, 70000)
    itime = int(a["itime"])

    def alive(d, p):
        m = d[p + "i"] == itime
        rows = np.stack([d[p + "x"][m], d[p + "y"][m], d[p + "z"][m], d[p + "m"].reshape(-1, d[p + "i"].size)[0][m]], axis=1)
        return rows[np.lexsort(rows.T[::-1])]
    before = np.concatenate([alive(a, "b"), alive(b, "b")])
    after = np.concatenate([alive(a, "a"), alive(b, "a")])
    before, after = (v[np.lexsort(v.T[::-1])] for v in (before, after))
    assert before.shape == after.shape and np.array_equal(before, after)
    na, nb = int((a["ai"] == itime).sum()), int((b["ai"] == itime).sum())
    nb0 = int((b["bi"] == itime).sum())
    assert nb > nb0 and nb - nb0 <= 70000 and int(a["n_after"]) == 260000 - 70000
    assert int(a["due"]) == na and int(b["due"]) == nb                         # every particle, received ones included, advanced


def test_bench_two_ranks_is_the_single_rank_job(built, tmp_path):
    """bench.py --gpus 2 as the driver will start it (here: bench.py launches its two ranks itself, as fresh child processes;
    on this one-GPU box they share the device and reduce over the host transport -- on a node each has its own GPU and RCCL):
    the line reports two ranks, every particle of the one cloud is alive on some rank, and the all-reduced concentration
    grid is the single-rank run's (mpi_mod.f90:2451-2492, timemanager_mpi.f90:552-562)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    lines = {}
    for n in (1, 2):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--config", "4", "--particles", "2e6", "--steps", "2", "--warmup", "1",
               "--no-cpu-baseline", "--no-pmc"]
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(tmp_path))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines[n] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    one, two = lines[1], lines[2]
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["live_particles_all_ranks"] == one["config"]["live_particles_all_ranks"] == 2000000
    assert two["config"]["particles_per_gpu"] == 1000000 and two["config"]["reduction_transport"] is not None
    assert two["config"]["time_blended_packs"] == one["config"]["time_blended_packs"]
    a, b = one["config"]["gridunc_sum_all_ranks"], two["config"]["gridunc_sum_all_ranks"]
    assert a > 0 and abs(a - b) <= 1e-6 * a, (a, b)
    assert abs(two["config"]["particle_steps_timed"] - one["config"]["particle_steps_timed"]) <= 2
