"""Convective mixing of particles: convmix / calcmatrix / convect43c / redist.

CPU (-m "not gpu"): the C restatement oracle/convect_oracle.c against the committed fixtures tests/golden/conv_*.npz, which
hold what the flang build of the unmodified CONVECT / TLIFT, redist, sort2, f_qvsat, ew, ran3 left after three passes
(generator: tests/golden/make_golden_conv.py), and against the live build where oracle/_ref exists -- bit for bit, r4 and r8.
calcmatrix.f90 and convmix.f90 need ecCodes and cannot be compiled here: their 160 lines of glue are restated in the driver and
in the oracle alike (parity unpinned for those two).
GPU (-m gpu): fpx_convmix through the C ABI against the oracle."""
import os

import numpy as np
import pytest

from flexpart_amd import synthetic as syn
from oracle import scenario_io as sio
from oracle.oracle import conv_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {"forward": dict(), "backward": dict(ldirect=-1, seed=23), "nested": dict(nest=True, seed=31)}
KEYS = ("ztra1", "cbaseflux", "lconv", "nconvtop", "fm_col", "fmassfrac")


@pytest.mark.parametrize("kind", ["r8", "r4"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_convection_fixtures(name, kind):
    cs = syn.convection_case(**CASES[name])
    gold = np.load(os.path.join(GOLD, f"conv_{name}_{kind}.npz"))
    calls = conv_oracle(cs, kind)
    assert len(calls) == int(gold["ncalls"])
    z0 = np.asarray(cs["ztra1"], dtype=np.float32 if kind == "r4" else np.float64).astype(np.float64)
    for i, c in enumerate(calls):
        for k in KEYS + (("cbasefluxn",) if "nest" in cs else ()):
            assert np.array_equal(np.asarray(c[k]), gold[f"c{i}_{k}"]), (name, kind, i, k)
    if "nest" in cs:      # particles inside the nest take the nest's soundings and its own mass-flux field
        g = cs["nestgeom"]
        inn = (cs["xtra1"] > g[0] + 0.01) & (cs["xtra1"] < g[2] - 0.01) & (cs["ytra1"] > g[1] + 0.01) & (cs["ytra1"] < g[3] - 0.01)
        assert inn.sum() > 500 and (calls[0]["ztra1"] != z0)[inn].sum() > 100
        assert not np.array_equal(calls[0]["cbasefluxn"], np.asarray(cs["cbasefluxn"]))
    # the scenario does what it is there for: a quarter of the visited columns convect, deep ones reach the top of the matrix,
    # the cloud-base mass flux relaxes from call to call, several hundred particles are displaced per call, dozens of them by kilometres
    first, last = calls[0], calls[-1]
    visited = first["lconv"] >= 0
    assert 60 < (first["lconv"] == 1).sum() < 0.5 * visited.sum() and first["nconvtop"].max() >= int(cs["nconvlev"])
    assert (first["lconv"] == 0).sum() > 100
    assert not np.array_equal(first["cbaseflux"], last["cbaseflux"]) and first["cbaseflux"].max() > 0.05
    moved = first["ztra1"] != z0
    assert moved.sum() > 500 and (np.abs(first["ztra1"] - z0)[moved] > 1000.0).sum() > 10
    assert (first["rn"] >= 0).sum() >= moved.sum() and not moved[~np.asarray(cs["due"])[:, 0]].any()


@pytest.mark.ref
@pytest.mark.skipif(not sio.have_conv_ref("r8"), reason="flang-built reference not present (GPU box)")
@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_oracle_matches_live_convection_reference(kind):
    cs = syn.convection_case(nx=20, ny=12, nuvz=60, n=2500, ncalls=2, seed=5)
    ref = sio.run_conv_reference(cs, kind)
    orc = conv_oracle(cs, kind)
    for i, (r, o) in enumerate(zip(ref, orc)):
        for k in KEYS:
            assert np.array_equal(np.asarray(r[k]), np.asarray(o[k])), (kind, i, k)


# ---------------------------------------------------------------------------------------------------------
# GPU: fpx_convmix through the C ABI
# ---------------------------------------------------------------------------------------------------------
def _engine(cs, kind, rng_mode, seed=3):
    from flexpart_amd.engine import Engine
    rb = 8 if kind == "r8" else 4
    nx, ny, nuvz = (int(v) for v in cs["grid"])
    n = int(cs["npart"])
    sc = syn.small(n=n, nx=nx, ny=ny, nz=30, nsteps=1, global_grid=False, ldirect=int(cs["ldirect"]))
    sc["xtra1"], sc["ytra1"], sc["ztra1"] = cs["xtra1"], cs["ytra1"], cs["ztra1"]
    if "nest" in cs:      # the same nested wind field for the trajectory part of the engine (fpx_nests_init)
        g = cs["nestgeom"]
        syn.add_nest(sc, int(g[0]), int(g[1]), int(g[2]), int(g[3]), factor=int(g[4]))
        assert tuple(sc["nest"]) == tuple(cs["nest"])
    eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=rng_mode, seed=seed)
    eng.set_windtime(cs["memtime"], (1, 2))
    eng.conv_init(cs)
    eng.cbaseflux(cs["cbaseflux"])
    for slot in (1, 2):
        eng.upload_conv_fields(slot, *(np.asarray(cs[k])[slot - 1] for k in ("ps", "tt2", "td2", "tth", "qvh")))
        if "nest" in cs:
            eng.upload_conv_nest_fields(1, slot, *(np.asarray(cs[k + "n"])[slot - 1] for k in ("ps", "tt2", "td2", "tth", "qvh")))
    if "nest" in cs:
        eng.cbaseflux_nest(1, np.asarray(cs["cbasefluxn"]).shape, cs["cbasefluxn"])
    return eng, sc


def _call(eng, sc, cs, ic, z):
    itime = int(cs["itimes"][ic])
    p = dict(sc, ztra1=z, itra1=np.where(np.asarray(cs["due"])[:, ic], itime, itime + 12345).astype(np.int32))
    eng.upload_particles_from_scenario(p)
    moved = eng.convmix(itime)
    return moved, eng.download()["ztra1"].astype(np.float64)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["r8", "r4"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_device_convmix_matches_the_oracle(built, name, kind):
    """fpx_convmix in the parity mode (serial ran3 stream replayed in the reference's sort2 order) against the oracle, three
    consecutive calls: the cloud-base mass flux of every column to rounding (it decides which columns convect), the particle
    heights -- nearly all of them to rounding; a particle may land in another level where a matrix entry differs in the last
    bits (the device's exp / log / pow are not glibc's) and its random number lies within that of a threshold."""
    from flexpart_amd.engine import RNG_TABLE_SEQ
    cs = syn.convection_case(**CASES[name])
    eng, sc = _engine(cs, kind, RNG_TABLE_SEQ)
    cs = dict(cs, height_nz=float(np.asarray(sc["height"])[-1]))
    want = conv_oracle(cs, kind)
    tol = 1e-9 if kind == "r8" else 2e-4
    z = np.asarray(cs["ztra1"], dtype=np.float64)
    for ic, w in enumerate(want):
        moved, z = _call(eng, sc, cs, ic, z)
        cb = eng.cbaseflux()
        assert np.abs(cb - w["cbaseflux"]).max() <= tol * np.abs(w["cbaseflux"]).max(), (name, kind, ic)
        assert np.array_equal(cb > 0, w["cbaseflux"] > 0)
        if "nest" in cs:
            cbn = eng.cbaseflux_nest(1, w["cbasefluxn"].shape)
            assert np.abs(cbn - w["cbasefluxn"]).max() <= tol * np.abs(w["cbasefluxn"]).max(), (name, kind, ic)
        close = np.abs(z - w["ztra1"]) <= tol * np.maximum(np.abs(w["ztra1"]), 1.0)
        assert close.mean() >= 0.995, (name, kind, ic, close.mean())
        assert moved >= (w["rn"] >= 0).sum() > 500
        z = w["ztra1"].copy()                       # continue from the oracle's state: each call is compared on equal input
    eng.close()


def _edge_cases():
    base = syn.convection_case(n=1200, ncalls=1, seed=41)
    n = int(base["npart"])
    out = {}
    out["nothing_due"] = dict(base, due=np.zeros((n, 1), bool))
    cold = dict(base)                                              # no column gets past CONVECT's early exits; a mass flux from before
    cold["tth"] = np.asarray(base["tth"]) - 45.0
    cold["tt2"] = np.asarray(base["tt2"]) - 45.0
    cold["td2"] = np.asarray(base["td2"]) - 60.0
    cold["qvh"] = np.asarray(base["qvh"]) * 1e-3
    out["no_column_convects"] = cold
    drew = int(np.flatnonzero(conv_oracle(dict(base, height_nz=30000.0), "r8")[0]["rn"] >= 0)[0])      # a particle that is lifted
    out["one_particle"] = dict(base, npart=1, due=np.ones((1, 1), bool), **{k: np.asarray(base[k])[drew:drew + 1] for k in ("xtra1", "ytra1", "ztra1")})
    high = dict(base, ztra1=np.where(np.arange(n) % 3 == 0, 29999.9, np.asarray(base["ztra1"])))     # a third of them at the model top
    out["particles_at_the_top"] = high
    corner = dict(base, xtra1=np.where(np.arange(n) % 2 == 0, 0.2, float(base["grid"][0]) - 1.2),          # all in the outermost columns
                  ytra1=np.where(np.arange(n) % 4 < 2, 0.2, float(base["grid"][1]) - 1.2))
    out["outermost_columns"] = corner
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_device_convmix_edge_cases(built, kind):
    """No particle due; no column that gets past CONVECT's early exits (the mass fluxes of before are still relaxed or reset); a
    single particle; particles at the top of the model (the clamp below height(nz)); all particles in the four outermost
    columns -- against the oracle, in parity mode."""
    from flexpart_amd.engine import RNG_TABLE_SEQ
    tol = 1e-9 if kind == "r8" else 2e-4
    for name, cs in _edge_cases().items():
        eng, sc = _engine(cs, kind, RNG_TABLE_SEQ)
        cs = dict(cs, height_nz=float(np.asarray(sc["height"])[-1]))
        w = conv_oracle(cs, kind)[0]
        z0 = np.asarray(cs["ztra1"], dtype=np.float64)
        moved, z = _call(eng, sc, cs, 0, z0)
        cb = eng.cbaseflux()
        eng.close()
        assert np.abs(cb - w["cbaseflux"]).max() <= tol * max(np.abs(w["cbaseflux"]).max(), 1e-30), (name, kind)
        assert np.array_equal(cb > 0, w["cbaseflux"] > 0), (name, kind)
        close = np.abs(z - w["ztra1"]) <= tol * np.maximum(np.abs(w["ztra1"]), 1.0)
        assert close.mean() >= (0.995 if len(z) > 1 else 1.0), (name, kind, close.mean())
        rt = np.float32 if kind == "r4" else np.float64
        z0r = z0.astype(rt).astype(np.float64)
        if name == "nothing_due":
            assert moved == 0 and np.array_equal(z, z0r) and np.array_equal(w["ztra1"], z0r)
            assert np.array_equal(cb, np.asarray(cs["cbaseflux"]).astype(rt).astype(np.float64))
        if name == "no_column_convects":
            assert (w["lconv"] == 1).sum() == 0 and np.array_equal(w["ztra1"], z0r) and np.array_equal(z, z0r)
        if name == "particles_at_the_top":
            top = (np.arange(len(z)) % 3 == 0) & np.asarray(cs["due"])[:, 0]
            conv_cols = w["lconv"].ravel()[np.rint(cs["ytra1"]).astype(int) * int(cs["grid"][0]) + np.rint(cs["xtra1"]).astype(int)] == 1
            assert (top & conv_cols).sum() > 20 and np.all(w["ztra1"][top & conv_cols] == float(cs["height_nz"]) - 0.5)
        if name == "outermost_columns":
            assert (w["lconv"] >= 0).sum() == 4


def _coupled_oracle(cs, sc, kind, ncalls):
    """timemanager's order on the restatements: convmix (convect_oracle.c), then the particle loop (flexpart_oracle.c), on the
    one ran3 stream random_mod gives both -- redist's seed -88 restarts it at the first call of convmix, advance's -7 at the
    first call of advance, from then on each continues where the other stopped."""
    from oracle.oracle import Oracle
    orc = Oracle(sc, kind)
    orc.lib.orc_set_parallel_semantics(orc.h, 1)
    stages = []

    def trajectory_step(ic, arr, words):
        stages.append(("convmix", orc.x.copy(), orc.y.copy(), arr["z"].copy()))
        orc.z[:] = arr["z"].astype(orc.rt)
        w = orc.ran3_words()
        w[:59] = words[:59]
        orc.set_ran3_words(w)
        orc.step()
        st = orc.state()
        stages.append(("step", st["xtra1"], st["ytra1"], st["ztra1"]))
        arr["x"][:], arr["y"][:], arr["z"][:] = st["xtra1"], st["ytra1"], st["ztra1"]
        words[:59] = orc.ran3_words()[:59]

    cso = dict(cs, due=np.ones((int(cs["npart"]), ncalls), bool), itimes=np.asarray(cs["itimes"])[:ncalls])
    want = conv_oracle(cso, kind, between=trajectory_step)
    return stages, want


def test_coupled_oracle_runs_on_one_stream():
    """CPU: the coupled restatements run, convection lifts particles, and the stream handed back and forth matters -- with the
    state not handed over the trajectory step draws other table positions."""
    cs = syn.convection_case(n=1500, ncalls=2)
    nx, ny, _ = (int(v) for v in cs["grid"])
    sc = syn.small(n=int(cs["npart"]), nx=nx, ny=ny, nz=30, nsteps=1, global_grid=False)
    sc["xtra1"], sc["ytra1"], sc["ztra1"] = cs["xtra1"], cs["ytra1"], cs["ztra1"]
    cs = dict(cs, height_nz=float(np.asarray(sc["height"])[-1]))
    stages, want = _coupled_oracle(cs, sc, "r8", 2)
    assert [s[0] for s in stages] == ["convmix", "step", "convmix", "step"]
    assert (want[0]["rn"] >= 0).sum() > 200 and (want[1]["rn"] >= 0).sum() > 200
    from oracle.oracle import Oracle
    alone = Oracle(sc, "r8")
    alone.lib.orc_set_parallel_semantics(alone.h, 1)
    alone.z[:] = stages[0][3]
    alone.step()                                   # first trajectory step: advance's own seed restarts the stream -- equal
    assert np.array_equal(alone.state()["ztra1"], stages[1][3])
    alone.z[:] = stages[2][3]
    alone.step()                                   # second: the coupled run continues behind redist's draws -- not equal
    assert (alone.state()["ztra1"] != stages[3][3]).mean() > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_convmix_and_the_particle_loop_share_the_serial_stream(built, kind):
    """Parity mode, timemanager's order: fpx_convmix, fpx_step, three times, against the coupled restatements.  redist and
    advance draw from one ran3 stream: were the engine's replay of either off by a single draw, every later table position
    of advance would differ and no PBL particle would agree."""
    from flexpart_amd.engine import RNG_TABLE_SEQ
    cs = syn.convection_case(n=3000)
    ncalls = len(cs["itimes"])
    eng, sc = _engine(cs, kind, RNG_TABLE_SEQ)
    cs = dict(cs, height_nz=float(np.asarray(sc["height"])[-1]))
    stages, want = _coupled_oracle(cs, sc, kind, ncalls)
    eng.upload_particles_from_scenario(sc)
    tol = 1e-9 if kind == "r8" else 2e-4
    got = []
    for ic in range(ncalls):
        moved = eng.convmix(eng.itime)
        assert moved >= (want[ic]["rn"] >= 0).sum() > 300
        d = eng.download()
        got.append((d["xtra1"], d["ytra1"], d["ztra1"].astype(np.float64)))
        eng.step()
        d = eng.download()
        got.append((d["xtra1"], d["ytra1"], d["ztra1"].astype(np.float64)))
    eng.close()
    ok = np.ones(int(cs["npart"]), bool)
    for (what, x, y, z), (gx, gy, gz) in zip(stages, got):
        ok &= (np.abs(gx - x) <= tol * max(np.abs(x).max(), 1.0)) & (np.abs(gy - y) <= tol * max(np.abs(y).max(), 1.0)) & \
              (np.abs(gz - z) <= tol * np.maximum(np.abs(z), 1.0))
    # a particle that lands in another level (last-bit differences of a matrix entry, see above) stays apart from then on
    assert ok.mean() >= (0.985 if kind == "r8" else 0.95), ok.mean()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["r8", "r4"])
@pytest.mark.parametrize("name", ["forward", "backward", "nested"])
def test_level_parallel_kernels_equal_the_one_lane_kernel(built, name, kind):
    """The mixing computation with a lane per (column, level) -- k_conv_prelude / rows / cols / flux / matrix -- forms every
    sum in the order of the one-lane-per-column kernel (fpx_set_option "conv_one_lane"): heights and cloud-base mass fluxes bit for bit,
    over three calls (the mass flux of one call is the input of the next)."""
    import os
    from flexpart_amd.engine import RNG_PHILOX
    cs = syn.convection_case(**CASES[name])
    got = {}
    for mode in ("levels", "one_lane"):
        try:
            eng, sc = _engine(cs, kind, RNG_PHILOX)
            if mode == "one_lane":
                eng.set_option("conv_one_lane", 1)
            z = np.asarray(cs["ztra1"], dtype=np.float64)
            out = []
            for ic in range(len(cs["itimes"])):
                moved, z = _call(eng, sc, cs, ic, z)
                out.append((moved, z.copy(), eng.cbaseflux().copy()))
                if "nest" in cs:
                    out.append(eng.cbaseflux_nest(1, np.asarray(cs["cbasefluxn"]).shape).copy())
            eng.close()
        finally:
            pass
        got[mode] = out
    assert got["levels"][0][0] > 500
    for a, b in zip(got["levels"], got["one_lane"]):
        if isinstance(a, tuple):
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        else:
            assert np.array_equal(a, b)


@pytest.mark.gpu
def test_device_convmix_with_the_counter_generator(built):
    """Counter RNG: individual displacements differ from the serial stream's by construction; the ensemble must not.  Same
    columns convect, particles outside convective columns and particles that are not due stay where they are, the fraction
    that changes level and the mean displacement agree with the oracle's, results do not depend on the storage order, and a
    scratch budget that forces several column batches gives the same result as one batch."""
    from flexpart_amd.engine import RNG_PHILOX
    cs = syn.convection_case(n=40000, seed=19)
    eng, sc = _engine(cs, "r8", RNG_PHILOX)
    cs = dict(cs, height_nz=float(np.asarray(sc["height"])[-1]))
    want = conv_oracle(cs, "r8")[0]
    z0 = np.asarray(cs["ztra1"], dtype=np.float64)
    moved, z = _call(eng, sc, cs, 0, z0)
    cb = eng.cbaseflux()
    eng.close()
    assert np.abs(cb - want["cbaseflux"]).max() <= 1e-9 * want["cbaseflux"].max()
    due = np.asarray(cs["due"])[:, 0]
    col = np.rint(cs["ytra1"]).astype(int) * int(cs["grid"][0]) + np.rint(cs["xtra1"]).astype(int)
    conv = want["lconv"].ravel()[col] == 1
    assert np.array_equal(z[~due], z0[~due]) and np.array_equal(z[due & ~conv], z0[due & ~conv])
    jump_dev, jump_orc = np.abs(z - z0) > 500.0, np.abs(want["ztra1"] - z0) > 500.0
    assert jump_orc.sum() > 150 and abs(jump_dev.sum() - jump_orc.sum()) < 4.0 * np.sqrt(jump_orc.sum())
    assert abs(np.abs(z - z0)[jump_dev].mean() - np.abs(want["ztra1"] - z0)[jump_orc].mean()) < 0.25 * np.abs(want["ztra1"] - z0)[jump_orc].mean()
    small = ~jump_dev & ~jump_orc & due & conv             # mostly the compensating subsidence, which takes no random number
    assert small.sum() > 5000 and (np.abs(z - want["ztra1"])[small] < 1e-6).mean() > 0.9
    # storage order and batching
    import os
    eng, sc = _engine(cs, "r8", RNG_PHILOX)
    eng.upload_particles_from_scenario(dict(sc, ztra1=z0, itra1=np.where(due, 0, 12345).astype(np.int32)))
    eng.sort()
    eng.set_option("conv_scratch_mb", 8)
    eng.convmix(0)
    z2 = eng.download()["ztra1"].astype(np.float64)
    eng.close()
    assert np.array_equal(z2, z)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_fortran_host_convmix(built, kind):
    """The real Fortran host: convref_rK either runs the reference's CONVECT / redist around our calcmatrix / convmix glue, or
    hands com_mod / conv_mod (tth, qvh, ps, tt2, td2 with their nxmax strides, akz ... bkm, cbaseflux, the particle arrays) to
    flexgpu_conv_init / flexgpu_upload_conv_fields / flexgpu_convmix in the serial-stream parity mode and downloads the particles
    -- the same heights and mass fluxes, to the tolerances of the C-ABI test."""
    if not sio.have_conv_ref(kind):
        pytest.skip("oracle/_ref/convref binaries not present in this snapshot")
    cs = syn.convection_case(ncalls=2)
    ref = sio.run_conv_reference(cs, kind)
    gpu = sio.run_conv_reference(cs, kind, gpu=True)
    tol = 1e-9 if kind == "r8" else 2e-4
    # the second call starts from each run's own first result: compare it on the particles that agreed after the first
    agreed = np.ones(int(cs["npart"]), bool)
    for ic, (g, r) in enumerate(zip(gpu, ref)):
        assert np.abs(g["cbaseflux"] - r["cbaseflux"]).max() <= tol * r["cbaseflux"].max(), (kind, ic)
        close = np.abs(g["ztra1"] - r["ztra1"]) <= tol * np.maximum(np.abs(r["ztra1"]), 1.0)
        assert close[agreed].mean() >= 0.99, (kind, ic, close[agreed].mean())
        agreed &= close
    assert (ref[0]["ztra1"] != np.asarray(cs["ztra1"], dtype=np.float32 if kind == "r4" else np.float64)).sum() > 500
