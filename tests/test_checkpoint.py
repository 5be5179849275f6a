"""Lossless checkpoint (last clause): fpx_checkpoint_write / fpx_checkpoint_read.

The reference's own restart (partoutput -> readpartpositions) drops the turbulent state; this pair keeps everything the
particle loop carries, so a run continued from the file must be the uninterrupted run -- bit for bit for the particle
arrays, and to the rounding of the atomic sums for the sampling grids."""
import numpy as np
import pytest

from flexpart_amd import synthetic as syn

pytestmark = pytest.mark.gpu

KEYS = ("xtra1", "ytra1", "ztra1", "uap", "ucp", "uzp", "us", "vs", "ws", "idt", "itra1", "itramem", "cbt", "xmass1", "npoint", "nclass")


def scenario():
    sc = syn.small(n=5000, nx=48, ny=32, nz=36, nsteps=4, ctl=5.0, ifine=4, cblflag=1, nspec=2)
    sc.update(lsettling=1, drydep=1, drydepspec=np.array([1, 1], np.int32), density=np.array([2000.0, 1500.0]),
              dquer=np.array([8.0, 4.0]), vsetaver=np.array([-0.004, -0.002]), cunningham=np.array([1.02, 1.04]),
              decay=np.array([1.0e-6, 0.0]), xmass=np.array([1.0, 1.0]))
    syn.add_outgrid(sc)
    syn.add_wet(sc, gas=False)
    for k in ("wetdepspec", "weta_gas", "wetb_gas", "crain_aero", "csnow_aero", "ccn_aero", "in_aero", "henry"):
        sc[k] = np.repeat(np.asarray(sc[k]), 2)                      # both species are scavenged
    sc["dquer"] = np.array([8.0, 4.0])
    syn.add_outgrid_nest(sc)
    syn.add_receptors(sc)
    return sc


def advance(eng, nsteps, sort_after=None):
    for k in range(nsteps):
        if eng.itime != 0:
            eng.wetdepo()
        eng.step()
        eng.conccalc(eng.itime, 1.0)
        if sort_after == k:
            eng.sort()


def outputs(eng):
    g, d = eng.grids()
    gn, dn, wn = eng.grids_nest()
    return dict(gridunc=g, drygridunc=d, wetgridunc=eng.wetgrid(), griduncn=gn, drygriduncn=dn, wetgriduncn=wn, creceptor=eng.receptors())


@pytest.mark.parametrize("mode", ["philox", "table_seq"])
@pytest.mark.parametrize("rb", [8, 4])
def test_run_continued_from_a_checkpoint_is_the_uninterrupted_run(built, tmp_path, mode, rb):
    from flexpart_amd.engine import Engine, RNG_PHILOX, RNG_TABLE_SEQ
    rng = RNG_PHILOX if mode == "philox" else RNG_TABLE_SEQ
    sc = scenario()
    kw = dict(compute_real_bytes=rb, host_real_bytes=rb, rng_mode=rng, seed=77)
    a = Engine(sc, **kw)
    advance(a, 4, sort_after=1)
    want, want_out = a.download(), outputs(a)
    a.close()
    b = Engine(sc, **kw)
    advance(b, 2, sort_after=1)              # the checkpoint is taken from re-sorted storage spaces
    ck = tmp_path / "ckpt"
    b.checkpoint_write(ck, numparticlecount=4321)
    t_b = b.itime
    b.close()
    c = Engine(sc, **kw)                     # a fresh engine: same configuration, the scenario's particles are overwritten
    itime, n, npc = c.checkpoint_read(ck)
    assert (itime, n, npc) == (t_b, 5000, 4321)
    advance(c, 2)
    got, got_out = c.download(), outputs(c)
    c.close()
    for k in KEYS:
        assert np.array_equal(want[k], got[k]), k
    assert (want["itra1"] > 0).sum() > 4000 and np.abs(want["uzp"]).max() > 0
    for k, w in want_out.items():
        assert w.sum() > 0, k
        tol = 1e-12 if rb == 8 and k in ("gridunc", "griduncn", "creceptor") else 2e-5     # atomic sums in another order (f32 for deposition)
        assert np.abs(got_out[k] - w).max() <= tol * np.abs(w).max(), k


def test_the_reference_restart_is_lossy_where_the_checkpoint_is_not(built, tmp_path):
    """partoutput + readpartpositions (the reference's warm start) re-initialises the turbulent state: the continued run
    differs from the uninterrupted one; documented here next to the lossless pair."""
    from flexpart_amd.engine import Engine, RNG_PHILOX
    sc = syn.small(n=3000, nx=40, ny=24, nz=30, nsteps=3, ctl=5.0, ifine=4)
    kw = dict(compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_PHILOX, seed=5)
    a = Engine(sc, **kw)
    for _ in range(2):
        a.step()
    state_before = a.download()
    a.checkpoint_write(tmp_path / "ck")
    a.close()
    b = Engine(sc, **kw)
    b.checkpoint_read(tmp_path / "ck")
    state_after = b.download()
    b.close()
    for k in KEYS:
        assert np.array_equal(state_before[k], state_after[k]), k            # the file holds the state itself
    assert np.abs(state_after["uzp"]).max() > 0                               # ... including what a partposit dump omits


def test_checkpoint_of_another_configuration_is_refused(built, tmp_path):
    from flexpart_amd._lib import FpxError
    from flexpart_amd.engine import Engine, RNG_PHILOX, RNG_TABLE_SEQ
    sc = syn.small(n=500, nx=40, ny=24, nz=30, nsteps=1, ctl=5.0, ifine=4)
    a = Engine(sc, rng_mode=RNG_PHILOX, seed=5)
    a.step()
    a.checkpoint_write(tmp_path / "ck")
    a.close()
    for kw in (dict(rng_mode=RNG_TABLE_SEQ, seed=5), dict(rng_mode=RNG_PHILOX, seed=6), dict(rng_mode=RNG_PHILOX, seed=5, compute_real_bytes=4, host_real_bytes=4)):
        b = Engine(sc, **kw)
        before = b.download()
        with pytest.raises(FpxError):
            b.checkpoint_read(tmp_path / "ck")
        after = b.download()
        b.close()
        assert np.array_equal(before["xtra1"], after["xtra1"])
    (tmp_path / "junk").write_bytes(b"not a checkpoint")
    b = Engine(sc, rng_mode=RNG_PHILOX, seed=5)
    with pytest.raises(FpxError):
        b.checkpoint_read(tmp_path / "junk")
    b.close()


def test_truncated_checkpoint_and_checkpoint_of_another_grid_leave_the_engine_untouched(built, tmp_path):
    """The header carries the grid the arrays were sized with and the length of the whole file; both are checked before a
    single array is restored: a truncated file, a file with trailing bytes and a file of a run on another grid are refused
    with every particle where it was."""
    from flexpart_amd._lib import FpxError
    from flexpart_amd.engine import Engine, RNG_PHILOX
    sc = syn.small(n=700, nx=40, ny=24, nz=30, nsteps=1, ctl=5.0, ifine=4)
    a = Engine(sc, rng_mode=RNG_PHILOX, seed=5)
    a.step()
    a.checkpoint_write(tmp_path / "ck")
    want = a.download()
    a.close()
    raw = (tmp_path / "ck").read_bytes()
    (tmp_path / "short").write_bytes(raw[:len(raw) - 1000])
    (tmp_path / "long").write_bytes(raw + b"\0" * 8)
    b = Engine(sc, rng_mode=RNG_PHILOX, seed=5)
    before = b.download()
    for name, word in (("short", "truncated"), ("long", "truncated")):
        with pytest.raises(FpxError) as e:
            b.checkpoint_read(tmp_path / name)
        assert word in str(e.value) and "nothing was restored" in str(e.value)
        after = b.download()
        assert b.n == 700 and all(np.array_equal(before[k], after[k]) for k in KEYS)
    b.checkpoint_read(tmp_path / "ck")                       # the complete file still restores
    got = b.download()
    assert all(np.array_equal(want[k], got[k]) for k in KEYS)
    b.close()
    # same particle count, same byte count, another grid
    sc2 = syn.small(n=700, nx=24, ny=40, nz=30, nsteps=1, ctl=5.0, ifine=4)
    c = Engine(sc2, rng_mode=RNG_PHILOX, seed=5)
    with pytest.raises(FpxError) as e:
        c.checkpoint_read(tmp_path / "ck")
    assert "another grid" in str(e.value)
    c.close()
