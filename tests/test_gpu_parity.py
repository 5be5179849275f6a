"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical
seeded inputs, table-RNG mode (bit-faithful random stream).

Tolerance: BASELINE.json north_star -> trajectories within 1e-6 relative L-infinity.
For the all-fp64 build we assert far tighter (1e-9): the only differences are FMA
contraction and libm-vs-ocml rounding of exp/pow/erf.
"""
import numpy as np
import pytest

from flexpart_amd import synthetic as syn

pytestmark = pytest.mark.gpu

POS = ("xtra1", "ytra1", "ztra1")
VEL = ("uap", "ucp", "uzp", "us", "vs", "ws")


def run_pair(sc, kind="r8", nsteps=None, **ekw):
    from flexpart_amd.engine import Engine, RNG_TABLE_SEQ
    from oracle.oracle import Oracle
    rb = 8 if kind == "r8" else 4
    eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=RNG_TABLE_SEQ, **ekw)
    got = eng.run(nsteps)
    eng.close()
    orc = Oracle(sc, kind)
    orc.lib.orc_set_parallel_semantics(orc.h, 1)
    want = orc.run(nsteps)
    return got, want


def assert_close(got, want, tol_pos, tol_vel, max_diverged=0):
    """L-inf relative to each field's range.  A particle counts as *diverged* when its integer
    state differs or any component misses the tolerance (in f32 a 1-ulp libm difference can flip
    an int() truncation of the sub-step length, after which the particle consumes different
    random numbers -- SURVEY.md section 7 "precision").  At most `max_diverged` such particles
    are allowed (0 for the fp64 build); they are counted, never masked silently."""
    n = len(want["xtra1"])
    bad = np.zeros(n, bool)
    for k in ("idt", "itra1", "cbt"):
        bad |= np.asarray(got[k]) != np.asarray(want[k])
    worst = {}
    for keys, tol in ((POS, tol_pos), (VEL, tol_vel)):
        for k in keys:
            scale = max(np.abs(want[k]).max(), 1e-30)
            err = np.abs(got[k] - want[k]) / scale
            worst[k] = float(err[~bad].max()) if (~bad).any() else 0.0
            bad |= err > tol
    assert bad.sum() <= max_diverged, f"{bad.sum()} of {n} particles diverged; worst among the rest: {worst}"
    return int(bad.sum())


@pytest.mark.parametrize("name,kw", [
    ("hanna", dict(ctl=5.0, ifine=4)),
    ("hanna1_method0", dict(ctl=-5.0)),
    ("cbl", dict(ctl=5.0, ifine=4, cblflag=1)),
    ("above_pbl_only", dict(ctl=-5.0, hmix_const=100.0, frac_pbl=0.0, turb_off=True)),
])
def test_fp64_matches_oracle(built, name, kw):
    sc = syn.small(n=4000, nx=60, ny=40, nz=40, nsteps=4, **kw)
    got, want = run_pair(sc, "r8")
    for g, w in zip(got, want):
        assert_close(g, w, 1e-9, 1e-7)


@pytest.mark.parametrize("name,kw", [
    ("above_pbl_only", dict(ctl=-5.0, hmix_const=100.0, frac_pbl=0.0, turb_off=True)),
    ("hanna_no_mesoscale", dict(ctl=5.0, ifine=4, turb_off=True)),
    ("cbl", dict(ctl=5.0, ifine=4, cblflag=1)),
    ("backward_above_pbl", dict(ctl=-5.0, hmix_const=100.0, frac_pbl=0.0, turb_off=True, ldirect=-1)),
    ("backward_cbl", dict(ctl=5.0, ifine=4, cblflag=1, ldirect=-1)),
])
def test_time_blended_wind_packs_change_rounding_only(built, name, kw):
    """With fpx_config.blend_mode = 1 (automatic: from 3e7 particles of the whole run on) the step first blends the wind pack in time -- the weights are the same
    for every particle -- and the gathers that need no standard deviations (interpol_wind_short, and interpol_wind when the
    mesoscale term is off) read half the bytes: (y1*dt2 + y2*dt1)*dtt is then taken before the horizontal and vertical sums
    instead of after.  Forced on here: the results stay within the parity tolerance of the oracle and within rounding of the
    unblended engine."""
    from flexpart_amd.engine import Engine, RNG_TABLE_SEQ
    sc = syn.small(n=4000, nx=60, ny=40, nz=40, nsteps=4, **kw)
    got, want = run_pair(sc, "r8", blend_mode=1)
    for g, w in zip(got, want):
        assert_close(g, w, 1e-9, 1e-7)
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_TABLE_SEQ, blend_mode=2)
    plain = eng.run(4)
    assert eng.info("blended_steps") == 0 and eng.info("time_blended_packs") == 0
    eng.close()
    changed = 0.0
    for g, p in zip(got, plain):
        for k in POS:
            changed = max(changed, float(np.abs(g[k] - p[k]).max() / max(np.abs(p[k]).max(), 1e-30)))
            assert np.abs(g[k] - p[k]).max() <= 1e-11 * max(np.abs(p[k]).max(), 1e-30), k
    if "above_pbl" in name:
        assert changed > 0.0          # the blended path did run (it rounds differently)


def test_blend_switch_does_not_depend_on_timing_or_restarts(built, tmp_path):
    """The decision to blend the wind packs in time is a function of the configuration alone (round 3 read the length of an
    earlier step's work list from pinned memory without synchronisation): two runs under fpx_step_async are bitwise equal,
    every step of them blends from the first on, and a run continued from a checkpoint equals the uninterrupted one."""
    from flexpart_amd.engine import Engine, RNG_PHILOX
    sc = syn.small(n=20000, nx=60, ny=40, nz=40, nsteps=5, ctl=5.0, ifine=4, cblflag=1)
    kw = dict(rng_mode=RNG_PHILOX, seed=31, global_particles=50_000_000)
    keys = ("xtra1", "ytra1", "ztra1", "uap", "ucp", "uzp", "us", "vs", "ws", "idt", "itra1", "cbt")
    runs = []
    for _ in range(2):
        eng = Engine(sc, **kw)
        for _ in range(5):
            eng.step_async()
        eng.sync()
        assert eng.info("time_blended_packs") == 1 and eng.info("blended_steps") == 5
        runs.append(eng.download())
        eng.close()
    for k in keys:
        assert np.array_equal(runs[0][k], runs[1][k]), k
    eng = Engine(sc, **kw)
    for _ in range(2):
        eng.step()
    eng.checkpoint_write(tmp_path / "ck")
    eng.close()
    eng = Engine(sc, **kw)
    eng.checkpoint_read(tmp_path / "ck")
    for _ in range(3):
        eng.step()
    cont = eng.download()
    eng.close()
    for k in keys:
        assert np.array_equal(cont[k], runs[0][k]), k
    plain = Engine(sc, rng_mode=RNG_PHILOX, seed=31)          # 20000 particles in all: below the threshold, no blend
    assert plain.info("time_blended_packs") == 0
    plain.close()


@pytest.mark.parametrize("rb", [8, 4])
@pytest.mark.parametrize("name,kw", [
    ("cbl", dict(ctl=5.0, ifine=4, cblflag=1)),
    ("hanna_backward", dict(ctl=5.0, ifine=4, ldirect=-1)),
    ("aerosol", None),
    ("nest", None),
])
def test_time_slices_do_not_change_a_bit(built, name, kw, rb):
    """The Langevin kernel runs in time slices: a launch gives a particle a budget of passes, a particle that needs more is
    suspended into its hand-over record and continues in the next launch (k_pbl_loop).  Budgets of 1, 2 and 5 passes -- so
    that nearly every particle is suspended several times, on every path (CBL, Gaussian, settling + dry deposition, inside a
    nest) -- give every array of the single-launch run bit for bit, in the counter-RNG mode and with the serial stream; so do the
    two other reasons to suspend: a wave that moves on to another stability class, and a wave that is down to fewer than
    "pbl_drain_lanes" particles when the list is used up (64: every wave hands everything on as soon as the list is empty)."""
    from flexpart_amd.engine import Engine, RNG_PHILOX, RNG_TABLE_SEQ
    if kw is None:
        from test_oracle_cpu import golden_scenario
        sc = golden_scenario(name)
    else:
        sc = syn.small(n=3000, nx=40, ny=24, nz=30, nsteps=3, **kw)
    keys = ("xtra1", "ytra1", "ztra1", "uap", "ucp", "uzp", "us", "vs", "ws", "idt", "itra1", "cbt", "xmass1")
    for mode in (RNG_PHILOX, RNG_TABLE_SEQ):
        ref = None
        for slices, drain in (("0", 0), ("0,0,0", 32), ("0,0,0,0,0", 64), ("1,2,5,0", 0), ("3,3,3,3,3,3,3,3,3,3,3,3,3,3,0", 48), ("2,7,0", 16)):
            eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=mode, seed=77,
                         options={"pbl_slices": slices, "pbl_drain_lanes": drain})
            out = eng.run(3)
            assert eng.info("pbl_launches_per_step") == len(slices.split(","))
            eng.close()
            if ref is None:
                ref = out
                continue
            for a, b in zip(ref, out):
                for k in keys:
                    assert np.array_equal(a[k], b[k]), (mode, slices, k)


@pytest.mark.parametrize("name", ["polar", "aerosol", "cbl", "hanna", "nest", "nest_wet", "sampling", "sampling_nest", "backward", "backward_cbl",
                                  "limited_area", "three_species", "multi_release", "age_classes",
                                  "backward_drybkdep", "backward_drybkdep_nest", "backward_wetbkdep",
                                  "turboff", "interpolhmix", "domainfill", "quasilag_step", "nokernel"])
def test_fp64_matches_oracle_golden_scenarios(built, name):
    """The scenarios the golden fixtures were made on: polar caps through the stereographic maps
    (cmapf subset), an aerosol species with settling + dry deposition + decay, CBL, Hanna."""
    from test_oracle_cpu import golden_scenario
    sc = golden_scenario(name)
    got, want = run_pair(sc, "r8")
    for g, w in zip(got, want):
        assert_close(g, w, 1e-9, 1e-7)
        scale = np.abs(w["xmass1"]).max()
        assert np.abs(g["xmass1"] - w["xmass1"]).max() <= 1e-12 * scale


@pytest.mark.parametrize("rb", [8, 4])
@pytest.mark.parametrize("name", ["cbl", "polar", "nest", "aerosol"])
def test_particles_born_after_the_upload_are_initialised_at_their_time(built, name, rb):
    """timemanager.f90:553 initialises a particle in the step whose time equals its itramem.  A host may upload particles that are
    born later (itra1 = itramem = a future step): they are not due before, and the step at their time must run initialize() for
    them although nothing was uploaded or released since -- the engine finds the latest birth among the uploaded particles and
    keeps the k_prep instance with initialize() until then.  Against the oracle, and bit for bit against the same run with that
    instance at every step ("prep_init_always")."""
    from flexpart_amd.engine import Engine, RNG_TABLE_SEQ, RNG_PHILOX
    from oracle.oracle import Oracle
    from test_oracle_cpu import golden_scenario
    sc = golden_scenario(name)
    n = len(sc["xtra1"])
    dt = int(sc["lsynctime"]) * int(sc["ldirect"])
    born = int(sc["itime0"]) + dt * (np.arange(n) % 3)          # a third each at the first, second and third step
    sc["itra1"] = born.astype(np.int32)
    sc["itramem"] = born.astype(np.int32)
    kind = "r8" if rb == 8 else "r4"
    eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=RNG_TABLE_SEQ)
    got = eng.run(3)
    assert eng.counters()["n_initialized"] == n, "every particle is initialised exactly once, in its own step"
    eng.close()
    orc = Oracle(sc, kind)
    orc.lib.orc_set_parallel_semantics(orc.h, 1)
    want = orc.run(3)
    for g, w in zip(got, want):
        if rb == 8:
            assert_close(g, w, 1e-9, 1e-7)
        else:
            assert_close(g, w, 2e-6, 5e-3, max_diverged=int(0.02 * n))
    keys = ("xtra1", "ytra1", "ztra1", "uap", "ucp", "uzp", "us", "vs", "ws", "idt", "itra1", "cbt", "xmass1")
    for mode in (RNG_PHILOX, RNG_TABLE_SEQ):
        ref = None
        for always in (0, 1):
            eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=mode, seed=5, options={"prep_init_always": always})
            out = eng.run(3)
            eng.close()
            if ref is None:
                ref = out
                continue
            for a, b in zip(ref, out):
                for k in keys:
                    assert np.array_equal(a[k], b[k]), (mode, k)


@pytest.mark.parametrize("rb", [8, 4])
@pytest.mark.parametrize("case", ["settling_only", "drydep_only", "settling_only_polar"])
def test_settling_and_dry_deposition_are_switched_separately(built, case, rb):
    """k_prep / k_pbl_finish have two families of instances: one for runs with neither dry deposition nor settling (the settling
    routine is not compiled into it) and one with both compiled in, each still behind its run-time switch.  A run with only one
    of the two features takes the second family: settling without deposition velocities, deposition of a species that does not
    settle.  Both against the oracle; the polar variant runs the instance with the stereographic maps."""
    from test_oracle_cpu import golden_scenario
    sc = golden_scenario("polar" if case.endswith("polar") else "aerosol")
    if case.startswith("settling_only"):
        sc.update(lsettling=1, drydep=0, drydepspec=np.array([0], np.int32), density=np.array([2000.0]), dquer=np.array([8.0]),
                  vsetaver=np.array([-0.004]), cunningham=np.array([1.02]), decay=np.array([0.0]), xmass=np.array([1.0]))
        sc.pop("vdep", None)
    else:
        sc.update(lsettling=0)
    got, want = run_pair(sc, "r8" if rb == 8 else "r4")
    for g, w in zip(got, want):
        if rb == 8:
            assert_close(g, w, 1e-9, 1e-7)
            assert np.abs(g["xmass1"] - w["xmass1"]).max() <= 1e-12 * np.abs(w["xmass1"]).max()
        else:
            assert_close(g, w, 2e-6, 5e-3, max_diverged=int(0.02 * len(w["xtra1"])))
    # the feature that is on did act: settling moves the particles down against the same run without it / deposition takes mass
    if case.startswith("settling_only"):
        sc0 = dict(sc); sc0.update(lsettling=0)
        base, _ = run_pair(sc0, "r8" if rb == 8 else "r4")
        assert np.mean(got[-1]["ztra1"]) < np.mean(base[-1]["ztra1"])
    else:
        assert got[-1]["xmass1"].sum() < got[0]["xmass1"].sum() or got[-1]["xmass1"].sum() < float(np.asarray(sc["xmass1"]).sum())


@pytest.mark.parametrize("name", ["polar", "aerosol", "cbl", "hanna", "hanna1_method0", "above_pbl_only", "nest", "nest_wet", "sampling",
                                  "sampling_nest", "backward", "backward_cbl", "limited_area", "three_species", "multi_release", "age_classes",
                                  "backward_drybkdep", "backward_drybkdep_nest", "backward_wetbkdep",
                                  "turboff", "interpolhmix", "domainfill", "quasilag_step", "nokernel"])
def test_fp64_against_reference_fixtures(built, name):
    """HIP path directly against the outputs of the unmodified reference (tests/golden, flang r8 builds).  Only
    particles touched by the two order-dependent leaks of the serial code may differ, and WHICH
    particles those are is known: the oracle, run with the reference's serial semantics, records every particle
    that takes advance.f90:550 (D1) or is initialised with its predecessor's polar / lat-lon wind choice (D2).
    Every other particle must agree with the reference to 1e-9 in position and exactly in itra1."""
    import os
    from flexpart_amd.engine import Engine
    from oracle.oracle import Oracle
    from test_oracle_cpu import GOLD, golden_scenario
    sc = golden_scenario(name)
    gold = np.load(os.path.join(GOLD, f"{name}_r8.npz"))
    orc = Oracle(sc, "r8")
    leaks = orc.track_leaks()
    orc.run()
    affected = leaks != 0
    eng = Engine(sc)
    got = eng.run()
    eng.close()
    n = int(sc["npart"])
    bad = np.zeros(n, bool)
    for i, g in enumerate(got):
        for k in ("xtra1", "ytra1", "ztra1"):
            ref = gold[f"s{i}_{k}"]
            bad |= np.abs(g[k] - ref) > 1e-9 * np.abs(ref).max()
        bad |= g["itra1"] != gold[f"s{i}_itra1"]
    assert not (bad & ~affected).any(), f"{(bad & ~affected).sum()} of {n} particles outside the D1/D2 set differ from the reference"
    if name == "polar":
        assert affected.sum() > 100 and bad.sum() > 0      # the fixture does exercise D2
    else:
        assert affected.sum() <= 0.01 * n


@pytest.mark.parametrize("kind", ["r8", "r4"])
@pytest.mark.parametrize("name", ["backward_drybkdep", "backward_drybkdep_nest", "backward_wetbkdep"])
def test_backward_receptor_scavenging(built, name, kind):
    """Row a24: backward runs with DRYBKDEP / WETBKDEP (COMMAND ind_receptor 4 / 3).  The receptor block of
    timemanager.f90:564-598 runs once per particle before it is moved (get_vdep_prob.f90 -> interpol_vdep[_nests];
    get_wetscav.f90), sets xscav_frac1 and zeroes the mass of what is not scavenged; conccalc weights every contribution
    with max(xscav_frac1, 0) (conccalc.f90:177-181 ...).  Against the oracle (which reproduces the flang build of the
    unmodified routines, tests/golden/backward_*bkdep_*.npz) and against those fixtures directly: xscav_frac1 after the
    first step is a pure function of the release position, so it is compared for EVERY particle."""
    import os
    from flexpart_amd.engine import Engine, RNG_TABLE_SEQ
    from oracle.oracle import Oracle
    from test_oracle_cpu import GOLD, golden_scenario
    sc = golden_scenario(name)
    rb = 8 if kind == "r8" else 4
    eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=RNG_TABLE_SEQ)
    got = eng.run()          # wetdepo (mass loss only in a backward run) -> step -> conccalc, as the time manager orders them
    g, _ = eng.grids()
    eng.close()
    orc = Oracle(sc, kind)
    orc.lib.orc_set_parallel_semantics(orc.h, 1)
    want = orc.run()
    og, _ = orc.grids()
    n = int(sc["npart"])
    gold = np.load(os.path.join(GOLD, f"{name}_{kind}.npz"))
    # f32: the Laakso / Kyro polynomials of get_wetscav.f90:230-240 sum terms of 1e5 to an exponent of order 1, so one ulp of
    # log10f moves the scavenging coefficient by 1e-3 (the f32 wet-deposition grids are compared at 5e-3 for the same reason)
    tol = 1e-12 if kind == "r8" else (2e-3 if "wet" in name else 2e-6)
    for i, (gs, ws) in enumerate(zip(got, want)):
        for ref in (ws["xscav_frac1"], gold[f"s{i}_xscav_frac1"].astype(np.float64)):
            assert np.abs(gs["xscav_frac1"] - ref).max() <= tol * np.abs(ref).max(), (name, kind, i)
        assert not (gs["xscav_frac1"] < 0).any()
        # what is not scavenged at the receptor carries no mass from the first step on
        dead_mass = (ws["xscav_frac1"] == 0) & (ws["xmass1"] == 0)
        assert np.array_equal((gs["xmass1"] == 0) & (gs["xscav_frac1"] == 0), dead_mass)
        if kind == "r8":
            assert_close(gs, ws, 1e-9, 1e-7)
            assert np.abs(gs["xmass1"] - ws["xmass1"]).max() <= 1e-12 * np.abs(ws["xmass1"]).max()
        else:
            assert_close(gs, ws, 2e-6, 5e-3, max_diverged=int(0.02 * n))
    assert (got[0]["xscav_frac1"] > 0).sum() > 50 and (got[0]["xscav_frac1"] == 0).sum() > 50
    assert og.sum() > 0 and g.size == og.size
    g = g.reshape(og.shape)
    assert np.abs(g - og).max() <= (1e-12 if kind == "r8" else 5e-3) * og.max()
    assert abs(g.sum() - og.sum()) <= (1e-10 if kind == "r8" else 2e-3) * og.sum()


@pytest.mark.parametrize("kind", ["r8", "r4"])
@pytest.mark.parametrize("name", ["multi_release", "age_classes", "domainfill", "quasilag_step", "nokernel"])
def test_release_points_age_classes_and_terminations(built, name, kind):
    """domainfill / quasilag_step: the same cloud with mdomainfill = 1 (no settling, no mass-fraction test, nrelpointer = 1 in the
    grids although ioutputforeachrelease = 1) and with mquasilag = 1 (no mass-fraction test); nokernel: the host built with
    lusekerneloutput = .false. (par_mod.f90:39) -- every contribution into the particle's own cell, mother and nested grid.
    Per-release-point xmass / npart in the mass-fraction termination (timemanager.f90:663-666,681-686) and in the
    settling species pick (advance.f90:518-531), the maximum-age termination (:701-707), and the trailing indices
    (species, release point, uncertainty class, age class) of gridunc / drygridunc / wetgridunc and their nested twins
    (conccalc.f90:54-58,140-143; drydepokernel, wetdepokernel) -- against the oracle, which reproduces the flang
    builds of the unmodified reference on these scenarios (tests/golden/multi_release_*, age_classes_*)."""
    from flexpart_amd.engine import Engine, RNG_TABLE_SEQ
    from oracle.oracle import Oracle
    from test_oracle_cpu import golden_scenario
    sc = golden_scenario(name)
    rb = 8 if kind == "r8" else 4
    eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=RNG_TABLE_SEQ)
    got, stats = [], []
    for _ in range(int(sc["nsteps"])):
        if eng.itime != 0:
            eng.wetdepo()
        stats.append(eng.step())
        eng.conccalc(eng.itime, 1.0)
        got.append(eng.download())
    g, d = eng.grids()
    w = eng.wetgrid()
    gn = eng.grids_nest() if name in ("age_classes", "nokernel") else None
    eng.close()
    orc = Oracle(sc, kind)
    orc.lib.orc_set_parallel_semantics(orc.h, 1)
    want = orc.run()
    og, od = orc.grids()
    ow = orc.wetgrid()
    n = int(sc["npart"])
    for gs, ws in zip(got, want):
        if kind == "r8":
            assert_close(gs, ws, 1e-9, 1e-7)
            assert np.array_equal(gs["itra1"], ws["itra1"])
            assert np.abs(gs["xmass1"] - ws["xmass1"]).max() <= 1e-12 * np.abs(ws["xmass1"]).max()
        else:
            assert_close(gs, ws, 2e-6, 5e-3, max_diverged=int(0.02 * n))
            assert (gs["itra1"] != ws["itra1"]).sum() <= 0.02 * n
    dead = [int((ws["itra1"] == -999999999).sum()) for ws in want]
    kills = sum(s["n_min_mass"] + s["n_max_age"] + s["n_left_domain"] for s in stats)
    assert kills == int((got[-1]["itra1"] == -999999999).sum())
    if name != "nokernel":
        assert dead[0] > 0 and dead[-1] > dead[0]
        assert sum(s["n_max_age"] for s in stats) > 0
    if name in ("domainfill", "quasilag_step"):      # timemanager.f90:662-666: xmassfract = 1, nothing ends for want of mass
        assert sum(s["n_min_mass"] for s in stats) == 0
    if name == "multi_release":
        assert sum(s["n_min_mass"] for s in stats) > 0
        # release point 4 carries no mass at all: xmassfract stays 0, every one of its particles ends in the first epilogue
        p4 = np.asarray(sc["npoint"]) == 4
        assert p4.sum() > 0 and np.all(got[0]["itra1"][p4] == -999999999)
    tol_g = 1e-12 if kind == "r8" else 5e-3
    tol_d = 2e-5 if kind == "r8" else 5e-3
    if name == "nokernel":      # one age class, one uncertainty class, one release-point plane: the oracle drops those axes
        g, d, w = g.reshape(og.shape), d.reshape(od.shape), w.reshape(ow.shape)
        assert og.sum() > 0 and od.sum() > 0 and ow.sum() > 0
    assert g.shape == og.shape and d.shape == od.shape and w.shape == ow.shape and (g.ndim == 7 or name == "nokernel")
    assert np.abs(g - og).max() <= tol_g * og.max(), np.abs(g - og).max() / og.max()
    assert np.abs(d - od).max() <= tol_d * od.max(), np.abs(d - od).max() / od.max()
    assert np.abs(w - ow).max() <= tol_d * ow.max(), np.abs(w - ow).max() / ow.max()
    # plane by plane: mass that lands in the wrong (age, class, point) plane would leave the per-plane sums unequal
    for a, b, t8 in ((g, og, 1e-9), (d, od, 1e-5), (w, ow, 1e-5)) if g.ndim == 7 else ():      # the deposition grids are f32 sums in every build
        tail = tuple(range(3, a.ndim))
        assert np.abs(a.sum(axis=tail) - b.sum(axis=tail)).max() <= (t8 if kind == "r8" else 2e-2) * b.sum(axis=tail).max()
    if gn is not None:
        for a, b in zip(gn, orc.grids_nest()):
            a = a.reshape(b.shape) if name == "nokernel" else a
            assert a.shape == b.shape and b.sum() > 0
            assert np.abs(a - b).max() <= tol_d * b.max()


@pytest.mark.parametrize("name", ["nest", "nest_wet"])
def test_reference_typed_f32_on_nests(built, name):
    """BASELINE config 5 in its own precision: the f32 engine (reference typing: f32 state and fields, f64 xy) on a met
    nest (interpol_*_nests), with dry deposition and -- nest_wet -- wet deposition through the nest's own precipitation
    fields and the nested output grid.  Against the r4 oracle (which reproduces the r4 flang build of the nested
    reference variant, tests/golden/nest_r4.npz / nest_wet_r4.npz) and against those fixtures directly."""
    import os
    from test_oracle_cpu import GOLD, golden_scenario
    sc = golden_scenario(name)
    n = int(sc["npart"])
    got, want = run_pair(sc, "r4")
    div = 0
    for g, w in zip(got, want):
        div = assert_close(g, w, 2e-6, 5e-3, max_diverged=int(0.02 * n))
    # the nest is where a good part of the cloud is
    x, y = np.asarray(sc["xtra1"]), np.asarray(sc["ytra1"])
    nx, ny = int(sc["grid"][0]), int(sc["grid"][1])
    inside = (x > nx // 4) & (x < (2 * nx) // 3) & (y > ny // 4) & (y < (3 * ny) // 4)
    assert inside.sum() > 0.15 * n
    gold = np.load(os.path.join(GOLD, f"{name}_r4.npz"))
    bad = np.zeros(n, bool)
    for i, g in enumerate(got):
        for k in ("xtra1", "ytra1", "ztra1"):
            ref = gold[f"s{i}_{k}"].astype(np.float64)
            bad |= np.abs(g[k] - ref) > 2e-6 * np.abs(ref).max()
    assert bad.sum() <= 0.03 * n, f"{bad.sum()} of {n} particles differ from the r4 reference ({div} from the oracle)"
    m = gold[f"s{len(got) - 1}_xmass1"].astype(np.float64)
    assert np.abs(got[-1]["xmass1"] - m)[:, ~bad].max() <= 5e-3 * m.max()


@pytest.mark.parametrize("name,kw", [
    ("hanna", dict(ctl=5.0, ifine=4)),
    ("hanna1_method0", dict(ctl=-5.0)),
    ("cbl", dict(ctl=5.0, ifine=4, cblflag=1)),
])
def test_reference_typed_f32_matches_oracle(built, name, kw):
    """Reference typing (f32 state, f64 xy): rounding differs at 1e-7 per operation, so a few
    particles flip an int() truncation; they are counted, not hidden.  `cbl` runs the f32 instance of the Langevin
    kernel with the skewed CBL scheme (cbl.f90:70-210 with hardware rcp / sqrt / exp / log): the kernel instance of
    BASELINE config 5."""
    sc = syn.small(n=4000, nx=60, ny=40, nz=40, nsteps=3, **kw)
    got, want = run_pair(sc, "r4")
    assert_close(got[-1], want[-1], 2e-6, 5e-3, max_diverged=80)   # <= 2 % of 4000


@pytest.mark.parametrize("name", ["cbl", "backward_cbl"])
def test_reference_typed_f32_cbl_against_reference_fixtures(built, name):
    """The f32 CBL kernel instance on the golden scenarios `cbl` and `backward_cbl` (ldirect = -1): against the r4 oracle
    (same divergence accounting as the other f32 tests: <= 2 % of the particles may flip an int() truncation) and
    directly against the outputs of the r4 flang build of the unmodified reference (tests/golden/*_r4.npz)."""
    import os
    from test_oracle_cpu import GOLD, golden_scenario
    sc = golden_scenario(name)
    n = int(sc["npart"])
    got, want = run_pair(sc, "r4")
    div = 0
    for g, w in zip(got, want):
        div = assert_close(g, w, 2e-6, 5e-3, max_diverged=int(0.02 * n))
    gold = np.load(os.path.join(GOLD, f"{name}_r4.npz"))
    bad = np.zeros(n, bool)
    for i, g in enumerate(got):
        for k in ("xtra1", "ytra1", "ztra1"):
            ref = gold[f"s{i}_{k}"].astype(np.float64)
            bad |= np.abs(g[k] - ref) > 2e-6 * np.abs(ref).max()
    assert bad.sum() <= 0.03 * n, f"{bad.sum()} of {n} particles differ from the r4 reference ({div} from the oracle)"
    # the scheme was exercised: a good part of the cloud sits in columns with -h/L > 5 (cbl.f90 branch of advance.f90:405)
    assert np.abs(got[-1]["uzp"]).max() > 0


def test_padded_host_arrays(built):
    """nxmax/nymax/nzmax strides of the reference's static arrays are honoured."""
    sc = syn.small(n=1000, nx=40, ny=24, nz=30, nsteps=2)
    got, want = run_pair(sc, "r8", pad=(3, 2, 5))
    assert_close(got[-1], want[-1], 1e-9, 1e-7)


def test_rng_table_is_bit_identical(built):
    from flexpart_amd.engine import Engine
    from oracle.oracle import Oracle
    sc = syn.small(n=10, nx=40, ny=24, nz=30, nsteps=1)
    for kind, rb in (("r8", 8), ("r4", 4)):
        eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb)
        tab = eng.rannumb()
        eng.close()
        assert np.array_equal(tab, Oracle(sc, kind).rannumb())


def test_config1_closed_form(built):
    """Uniform wind, no turbulence: dx = u*dt*dxconst/cos(lat) per step (SURVEY 8c self-check i)."""
    from flexpart_amd.engine import Engine
    sc = syn.config1(n=1000, nsteps=2)
    eng = Engine(sc)
    out = eng.run()
    eng.close()
    g = sc["geom"]
    dxconst = 180.0 / (g[0] * syn.R_EARTH * syn.PI_REF)
    x0 = sc["xtra1"][0]
    step = 10.0 * 900.0 * dxconst / np.cos(20.0 * syn.PI_REF / 180.0)
    assert np.allclose(out[0]["xtra1"], x0 + step, rtol=0, atol=1e-9)
    assert np.allclose(out[1]["xtra1"], x0 + 2 * step, rtol=0, atol=1e-9)
    assert np.allclose(out[1]["ytra1"], sc["ytra1"], rtol=0, atol=1e-12)
    assert np.all(out[1]["itra1"] == 1800)


@pytest.mark.parametrize("gather", ["direct", "staged"])
@pytest.mark.parametrize("rb", [8, 4])
def test_locality_sort_does_not_change_results(built, gather, rb):
    """Sorting permutes device slots only: particle numbering at the boundary and every
    result (table RNG is indexed by particle number) must be unchanged -- bitwise.  Both gathers of the re-sort
    (direct per array; through one 128-byte record per particle, chosen for permutations without locality), two
    species with different masses, and particles uploaded by number after a sort."""
    from flexpart_amd.engine import Engine
    sc = syn.small(n=3000, nx=40, ny=24, nz=30, nsteps=3, ctl=5.0, ifine=4, nspec=2)
    sc["xmass1"] = np.stack([np.linspace(1.0, 2.0, 3000), np.linspace(5.0, 3.0, 3000)])
    a = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb)
    ra = a.run()
    a.close()
    b = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb, options={"permute": gather})
    b.sort()
    b.step()
    b.sort()
    b.step()
    b.step()
    rb_ = b.download()
    b.close()
    for k in ("xtra1", "ytra1", "ztra1", "uap", "ucp", "uzp", "us", "vs", "ws", "idt", "itra1", "cbt", "xmass1", "npoint", "nclass", "itramem"):
        assert np.array_equal(ra[-1][k], rb_[k]), k


def test_counter_rng_is_order_independent(built):
    """PHILOX mode: results depend on (seed, particle number, step) only, not on slot order."""
    from flexpart_amd.engine import Engine, RNG_PHILOX
    sc = syn.small(n=3000, nx=40, ny=24, nz=30, nsteps=2, ctl=5.0, ifine=4, cblflag=1)
    a = Engine(sc, rng_mode=RNG_PHILOX, seed=99)
    ra = a.run()
    a.close()
    b = Engine(sc, rng_mode=RNG_PHILOX, seed=99, sort_interval=1)
    b.sort()
    rb = b.run()
    b.close()
    for k in ("xtra1", "ytra1", "ztra1", "uzp", "idt", "itra1"):
        assert np.array_equal(ra[-1][k], rb[-1][k]), k
    c = Engine(sc, rng_mode=RNG_PHILOX, seed=100)
    rc = c.run()
    c.close()
    assert not np.array_equal(ra[-1]["ztra1"], rc[-1]["ztra1"])


@pytest.mark.parametrize("cblflag", [0, 1])
def test_counter_rng_matches_table_rng_statistically(built, cblflag):
    """The counter generator (Philox4x32-10 + clipped Box-Muller) replaces the reference's 1e6-entry Gaussian
    table and its shared ran3 stream; individual trajectories differ by construction, the ensemble must not.
    Same cloud, all three modes: the table modes (the reference's numbers: serial stream / per-particle start
    index) bound the sampling noise, the counter mode has to sit inside it -- mean and spread of the
    displacement and of the turbulent velocities after four synchronisation steps."""
    from flexpart_amd.engine import Engine, RNG_PHILOX, RNG_TABLE_SEQ, RNG_TABLE_COUNTER
    n = 120000
    sc = syn.small(n=n, nx=60, ny=40, nz=40, nsteps=4, ctl=5.0, ifine=4, cblflag=cblflag, frac_pbl=0.7)
    x0, y0 = np.asarray(sc["xtra1"]), np.asarray(sc["ytra1"])
    stats = {}
    for name, mode, seed in (("table_seq", RNG_TABLE_SEQ, 1), ("table_ctr", RNG_TABLE_COUNTER, 2), ("philox", RNG_PHILOX, 3)):
        eng = Engine(sc, compute_real_bytes=8, host_real_bytes=8, rng_mode=mode, seed=seed)
        r = eng.run()[-1]
        eng.close()
        alive = r["itra1"] != -999999999
        assert alive.mean() > 0.99
        dx, dy = (r["xtra1"] - x0)[alive], (r["ytra1"] - y0)[alive]
        dx = (dx + 29.5) % 59.0 - 29.5                      # cyclic domain
        q = dict(dx=dx, dy=dy, z=r["ztra1"][alive], up=r["uap"][alive], wp=r["uzp"][alive])
        stats[name] = {k: (v.mean(), v.std(), np.abs(v).mean()) for k, v in q.items()}
    for k in ("dx", "dy", "z", "up", "wp"):
        a, b, c = stats["table_seq"][k], stats["table_ctr"][k], stats["philox"][k]
        for j in range(3):
            scale = max(abs(a[1]), 1e-30)                   # the spread of the quantity sets the scale
            noise = 5.0 * scale / np.sqrt(n)                # 5 standard errors of a mean over n particles
            noise = max(noise, 3.0 * abs(a[j] - b[j]))      # and never tighter than table-vs-table
            assert abs(c[j] - a[j]) <= noise + 0.01 * scale, (k, j, a, b, c)


def test_fortran_host_drop_in_nests(built):
    """Same drop-in check for the nested-grid reference variant (par_mod_meteoswiss.f90: nxmax=721,
    maxnests=1): the shim hands the allocatable 5-D nest arrays uun, vvn, ... to the engine."""
    from oracle import scenario_io as sio
    if not sio.have_ref("r8n"):
        pytest.skip("oracle/_ref/flexref_r8n not present in this snapshot")
    from test_oracle_cpu import golden_scenario
    sc = golden_scenario("nest")
    ref = sio.run_reference(sc, "r8n")
    gpu = sio.run_reference(sc, "r8n", gpu=True, tag="gpun")
    n = int(sc["npart"])
    bad = np.zeros(n, bool)
    for a, b in zip(gpu["steps"], ref["steps"]):
        for k in ("xtra1", "ytra1", "ztra1"):
            bad |= np.abs(a[k] - b[k]) > 1e-9 * np.abs(b[k]).max()
    assert bad.sum() <= 0.01 * n, f"{bad.sum()} of {n} particles differ"


@pytest.mark.parametrize("case,kind", [("sampling_nest", "r8"), ("nest_wet", "r8n")])
def test_fortran_host_drop_in_sampling(built, case, kind):
    """Rows a21-a23 through the Fortran shim: the host's own gridunc/drygridunc/wetgridunc (+ the nested output
    grids and creceptor) are filled by flexgpu_conccalc / flexgpu_wetdepo / flexgpu_get_grids and compared with
    the reference's conccalc, drydepokernel[_nest], wetdepo, wetdepokernel[_nest] in the same binary.
    The D1/D2 order effects move a few particles by more than rounding, hence the 2 % bound."""
    from oracle import scenario_io as sio
    if not sio.have_ref(kind):
        pytest.skip("oracle/_ref binaries not present in this snapshot")
    from test_oracle_cpu import golden_scenario
    sc = golden_scenario(case)
    ref = sio.run_reference(sc, kind)
    gpu = sio.run_reference(sc, kind, gpu=True, tag="gpug")
    keys = ["gridunc", "drygridunc", "wetgridunc", "griduncn", "drygriduncn", "wetgriduncn"]
    if "creceptor" in ref:
        keys.append("creceptor")
    for k in keys:
        a, b = gpu[k], ref[k]
        assert a.shape == b.shape and b.sum() > 0, k
        assert np.abs(a - b).max() <= 0.02 * b.max(), (k, np.abs(a - b).max() / b.max())
        assert abs(a.sum() - b.sum()) <= 2e-3 * b.sum(), (k, a.sum(), b.sum())


@pytest.mark.parametrize("case", ["backward_drybkdep", "backward_wetbkdep"])
def test_fortran_host_backward_receptor_scavenging(built, case):
    """Row a24 through the Fortran shim: the host's DRYBKDEP / WETBKDEP switches, its xscav_frac1(maxpart,maxspec) and
    point_mod zpoint1/zpoint2 reach the engine through flexgpu_init / flexgpu_upload_particles; after every step the
    host's own xscav_frac1, xmass1 and -- at the end -- gridunc equal those of the reference's loop (timemanager.f90:564-598,
    conccalc.f90:177-181) in the same binary."""
    from oracle import scenario_io as sio
    if not sio.have_ref("r8"):
        pytest.skip("oracle/_ref binaries not present in this snapshot")
    from test_oracle_cpu import golden_scenario
    sc = golden_scenario(case)
    ref = sio.run_reference(sc, "r8")
    gpu = sio.run_reference(sc, "r8", gpu=True, tag="gpubk")
    n = int(sc["npart"])
    for i, (a, b) in enumerate(zip(gpu["steps"], ref["steps"])):
        assert np.abs(a["xscav_frac1"] - b["xscav_frac1"]).max() <= 1e-12 * np.abs(b["xscav_frac1"]).max(), (case, i)
        assert np.array_equal(a["xmass1"] == 0, b["xmass1"] == 0)
        bad = np.zeros(n, bool)
        for k in ("xtra1", "ytra1", "ztra1"):
            bad |= np.abs(a[k] - b[k]) > 1e-9 * np.abs(b[k]).max()
        assert bad.sum() <= 0.01 * n, f"{bad.sum()} of {n} particles differ"      # D1/D2 order effects of the serial host loop
    a, b = gpu["gridunc"], ref["gridunc"]
    assert a.shape == b.shape and b.sum() > 0
    assert np.abs(a - b).max() <= 0.02 * b.max() and abs(a.sum() - b.sum()) <= 2e-3 * b.sum()


@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_fortran_host_drop_in(built, kind):
    """The real Fortran host: oracle/_ref/flexref_rK holds the reference's com_mod arrays
    (static, nxmax/nymax/nzmax strides) and either runs the reference's own loop or hands the very
    same arrays to the engine through flexpart_amd/fortran/flexgpu_mod.f90 (ISO_C_BINDING).
    r8 host -> fp64 engine; r4 host (the reference as shipped) -> fp64 engine fed with f32 arrays."""
    from oracle import scenario_io as sio
    if not sio.have_ref(kind):
        pytest.skip("oracle/_ref binaries not present in this snapshot")
    sc = syn.small(n=2000, nx=60, ny=40, nz=40, nsteps=3, ctl=5.0, ifine=4, seed=4242)
    ref = sio.run_reference(sc, kind)
    gpu = sio.run_reference(sc, kind, gpu=True, tag="gpu")
    n = int(sc["npart"])
    bad = np.zeros(n, bool)
    tol = 1e-9 if kind == "r8" else 2e-5
    for a, b in zip(gpu["steps"], ref["steps"]):
        for k in ("xtra1", "ytra1", "ztra1"):
            bad |= np.abs(a[k] - b[k]) > tol * np.abs(b[k]).max()
    # D1 may touch a handful of PBL particles; r4: the engine computes in fp64 from
    # f32 inputs, so particles whose int() truncation flips diverge from the all-f32 reference
    limit = 0.01 * n if kind == "r8" else 0.05 * n
    assert bad.sum() <= limit, f"{bad.sum()} of {n} particles differ"
    assert np.array_equal(gpu["steps"][-1]["itra1"], ref["steps"][-1]["itra1"])


@pytest.mark.parametrize("variant,case", [("r8t", "turboff"), ("r4t", "turboff"), ("r8h", "interpolhmix")])
def test_fortran_host_passes_its_compile_time_switches(built, variant, case):
    """A host built with turboff = .true. or interpolhmix = .true. (com_mod.f90:777-778: compile-time parameters of the
    reference) hands them to the engine as run-time switches (flexgpu_init): the engine driven by that host reproduces that
    host's own loop -- which differs from the shipped behaviour (checked: the two fixtures are not the same trajectories)."""
    import os
    from oracle import scenario_io as sio
    from test_oracle_cpu import GOLD, golden_scenario
    if not sio.have_ref(variant):
        pytest.skip("oracle/_ref binaries not present in this snapshot")
    sc = golden_scenario(case)
    ref = sio.run_reference(sc, variant)
    gpu = sio.run_reference(sc, variant, gpu=True, tag="gpu")
    n = int(sc["npart"])
    bad = np.zeros(n, bool)
    tol = 1e-9 if variant.startswith("r8") else 2e-5
    for a, b in zip(gpu["steps"], ref["steps"]):
        for k in ("xtra1", "ytra1", "ztra1"):
            bad |= np.abs(a[k] - b[k]) > tol * np.abs(b[k]).max()
    assert bad.sum() <= (0.01 if variant.startswith("r8") else 0.05) * n, f"{bad.sum()} of {n} particles differ"
    stock = dict(sc)
    stock.pop(case)
    plain = sio.run_reference(stock, variant[:2])          # the shipped parameters on the same cloud
    moved = np.abs(plain["steps"][-1]["ztra1"] - ref["steps"][-1]["ztra1"]) > 1e-6 * np.abs(ref["steps"][-1]["ztra1"]).max()
    assert moved.sum() > 0.2 * n


@pytest.mark.parametrize("key,value", [("ipout", 3), ("iflux", 1), ("linit_cond", 1), ("linit_cond", 2)])
def test_in_loop_diagnostics_are_refused_not_dropped(built, key, value):
    """The block timemanager.f90:531-712 also calls partpos_average (ipout = 3, :617), calcfluxes (iflux = 1, :623) and
    initial_cond_calc (linit_cond >= 1, :631,702).  The engine does not compute them (out of scope); a host
    that asks for them gets FPX_ERR_UNSUPPORTED at fpx_create instead of a run that silently lacks their output."""
    from flexpart_amd.engine import Engine
    sc = syn.small(n=10, nx=20, ny=12, nz=10, nsteps=1)
    sc[key] = value
    with pytest.raises(Exception, match="not computed by this engine"):
        Engine(sc)
    sc[key] = 0 if key != "ipout" else 2
    Engine(sc).close()


@pytest.mark.parametrize("cbl", [0, 1])
def test_fortran_host_drives_the_f32_engine(built, cbl):
    """The reference as shipped (default real = 4 bytes) handing its com_mod arrays to the reference-typed f32 engine
    through flexgpu_mod (flexgpu_init(compute_real_bytes = 4)): the same arithmetic types on both sides, so the
    comparison is the f32 parity test (<= 2 % of the particles may flip an int() truncation), with and without the
    CBL scheme (cblflag = 1: the kernel instance of BASELINE config 5)."""
    from oracle import scenario_io as sio
    if not sio.have_ref("r4"):
        pytest.skip("oracle/_ref binaries not present in this snapshot")
    sc = syn.small(n=2000, nx=60, ny=40, nz=40, nsteps=3, ctl=5.0, ifine=4, seed=4242, cblflag=cbl)
    ref = sio.run_reference(sc, "r4")
    gpu = sio.run_reference(sc, "r4", gpu=32, tag="gpu32")
    n = int(sc["npart"])
    bad = np.zeros(n, bool)
    for a, b in zip(gpu["steps"], ref["steps"]):
        for k in ("xtra1", "ytra1", "ztra1"):
            bad |= np.abs(a[k] - b[k]) > 2e-6 * np.abs(b[k]).max()
    assert bad.sum() <= 0.03 * n, f"{bad.sum()} of {n} particles differ"     # 2 % truncation flips + the D1/D2 leaks of the serial host
    assert (gpu["steps"][-1]["itra1"] != ref["steps"][-1]["itra1"]).sum() <= 0.01 * n


@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_conccalc_and_dry_deposition_grids(built, kind):
    """gridunc (conccalc.f90: direct cell + 4-cell kernel, density interpolation ind_samp=-1) and
    drygridunc (drydepokernel.f90 from the step's epilogue) against the oracle.  Float atomics
    sum in a different order than the serial loop: tolerance relative to the largest cell."""
    from flexpart_amd.engine import Engine
    from oracle.oracle import Oracle
    from test_oracle_cpu import golden_scenario
    sc = syn.add_outgrid(golden_scenario("aerosol"))
    rb = 8 if kind == "r8" else 4
    eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb)
    eng.run()
    g, d = eng.grids()
    eng.close()
    orc = Oracle(sc, kind)
    orc.lib.orc_set_parallel_semantics(orc.h, 1)
    orc.run()
    og, od = orc.grids()
    g = g[0, 0, 0]; d = d[0, 0, 0]
    assert og.sum() > 0 and od.sum() > 0 and (og != 0).sum() > 500
    tol_g = 1e-12 if kind == "r8" else 2e-3     # f32: a few particles land in another cell
    tol_d = 2e-5 if kind == "r8" else 5e-3
    assert np.abs(g - og).max() <= tol_g * og.max(), np.abs(g - og).max() / og.max()
    assert np.abs(d - od).max() <= tol_d * od.max(), np.abs(d - od).max() / od.max()
    assert abs(g.sum() - og.sum()) <= 1e-6 * og.sum()
    # mass budget of the species: what left the particles by dry deposition + decay went somewhere


def test_point_release_hits_few_cells(built):
    """All particles in one cell: the wave-level pre-reduction path of the scatter-add."""
    from flexpart_amd.engine import Engine
    from oracle.oracle import Oracle
    sc = syn.config1(n=5000, nsteps=2)
    syn.add_outgrid(sc, nxg=85, nyg=65, nzg=4, outlon0=-10.0, outlat0=5.0, dxout=0.5, dyout=0.5,
                    ind_samp=0, old_fraction=0.0)
    eng = Engine(sc)
    eng.run()
    g, _ = eng.grids()
    eng.close()
    orc = Oracle(sc, "r8")
    orc.run()
    og, _ = orc.grids()
    assert (og != 0).sum() <= 4
    assert np.abs(g[0, 0, 0] - og).max() <= 1e-12 * og.max()
    assert abs(g.sum() - 2.0) < 1e-9        # 2 samples x total mass 1


def test_rccl_communicator_single_rank(built):
    """fpx_comm_unique_id / fpx_comm_init / all-reduce path with a one-rank communicator (the
    round driver exercises N = 2, 4, 8 on a full node)."""
    from flexpart_amd.engine import Engine
    sc = syn.add_outgrid(syn.small(n=500, nx=30, ny=20, nz=16, nsteps=1), nxg=12, nyg=8, nzg=3)
    eng = Engine(sc)
    eng.comm_init(eng.comm_unique_id(), 1, 0)
    eng.run()
    g1, _ = eng.grids(allreduce=True)
    g2, _ = eng.grids(allreduce=False, clear=True)
    g3, _ = eng.grids()
    eng.close()
    assert g1.sum() > 0 and np.array_equal(g1, g2) and g3.sum() == 0


@pytest.mark.parametrize("gas", [False, True])
@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_wet_deposition(built, kind, gas):
    """wetdepo.f90 + get_wetscav.f90 + interpol_rain.f90 + wetdepokernel.f90: particle masses after
    scavenging and the accumulated wetgridunc against the oracle (aerosol: below-cloud Laakso/Kyro
    polynomials + in-cloud CCN/IN activation; gas: power law below cloud + Henry in cloud)."""
    from flexpart_amd.engine import Engine
    from oracle.oracle import Oracle
    sc = syn.small(n=3000, nx=60, ny=40, nz=40, nsteps=3, ctl=5.0, ifine=4)
    sc.update(decay=np.array([1.0e-6]), xmass=np.array([1.0]))
    syn.add_wet(syn.add_outgrid(sc), gas=gas)
    rb = 8 if kind == "r8" else 4
    eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb)
    got = eng.run()
    w = eng.wetgrid()[0, 0, 0]
    eng.close()
    orc = Oracle(sc, kind)
    orc.lib.orc_set_parallel_semantics(orc.h, 1)
    want = orc.run()
    ow = orc.wetgrid()
    assert ow.sum() > 1.0 and want[-1]["xmass1"].min() < 0.9
    n = int(sc["npart"])
    # f32: the below-cloud aerosol coefficient is 10**(sum of terms ~1e5/log10(d)**k): one ulp of
    # log10f moves the result by ~1e-4, so the reference-typed build is only checked to 2e-3
    tol = 1e-11 if kind == "r8" else 2e-3
    bad = np.abs(got[-1]["xmass1"] - want[-1]["xmass1"]).ravel() > tol
    assert bad.sum() <= (0 if kind == "r8" else 0.02 * n), bad.sum()
    assert np.abs(w - ow).max() <= (2e-5 if kind == "r8" else 5e-3) * ow.max()
    assert abs(w.sum() - ow.sum()) <= (1e-5 if kind == "r8" else 1e-3) * ow.sum()


@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_nested_output_grid_and_receptors(built, kind):
    """conccalc.f90:301-441 (griduncn), drydepokernel_nest.f90, wetdepokernel_nest.f90 (floor(), always the
    kernel) and the receptor kernel conccalc.f90:451-498 against the oracle, which reproduces the reference's
    griduncn/drygriduncn/wetgriduncn/creceptor bit for bit (tests/golden/sampling_nest_*.npz)."""
    from flexpart_amd.engine import Engine
    from oracle.oracle import Oracle
    from test_oracle_cpu import golden_scenario
    sc = golden_scenario("sampling_nest")
    rb = 8 if kind == "r8" else 4
    eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb)
    eng.run()
    gn, dn, wn = (a[0, 0, 0] for a in eng.grids_nest())
    g, d = (a[0, 0, 0] for a in eng.grids())
    rec = eng.receptors()
    eng.close()
    orc = Oracle(sc, kind)
    orc.lib.orc_set_parallel_semantics(orc.h, 1)
    orc.run()
    og, od = orc.grids()
    ogn, odn, own = orc.grids_nest()
    orec = orc.receptors()
    assert ogn.sum() > 0 and odn.sum() > 0 and own.sum() > 0 and orec.max() > 0
    tol_g = 1e-12 if kind == "r8" else 2e-3
    tol_d = 2e-5 if kind == "r8" else 5e-3
    assert np.abs(g - og).max() <= tol_g * og.max()
    assert np.abs(gn - ogn).max() <= tol_g * ogn.max(), np.abs(gn - ogn).max() / ogn.max()
    assert np.abs(dn - odn).max() <= tol_d * odn.max(), np.abs(dn - odn).max() / odn.max()
    assert np.abs(wn - own).max() <= tol_d * own.max(), np.abs(wn - own).max() / own.max()
    assert np.abs(rec - orec).max() <= (1e-11 if kind == "r8" else 2e-3) * orec.max(), (rec, orec)   # f32: a few trajectories differ


def test_wet_deposition_on_nested_grid(built):
    """A particle inside a met nest is scavenged with the nest's own precipitation, cloud and temperature
    fields (get_wetscav.f90:82-101,126-128,150-151,197-199; interpol_rain_nests.f90); the deposit also goes to the
    nested output grid.  The oracle matches the reference (r8n build) bit for bit: tests/golden/nest_wet_r8.npz."""
    from flexpart_amd.engine import Engine
    from oracle.oracle import Oracle
    from test_oracle_cpu import golden_scenario
    sc = golden_scenario("nest_wet")
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=8)
    got = eng.run()
    w = eng.wetgrid()[0, 0, 0]
    gn, dn, wn = (a[0, 0, 0] for a in eng.grids_nest())
    eng.close()
    orc = Oracle(sc, "r8")
    orc.lib.orc_set_parallel_semantics(orc.h, 1)
    want = orc.run()
    ow = orc.wetgrid()
    ogn, odn, own = orc.grids_nest()
    assert ow.sum() > 1.0 and own.sum() > 1.0 and want[-1]["xmass1"].min() < 0.9
    bad = np.abs(got[-1]["xmass1"] - want[-1]["xmass1"]).ravel() > 1e-11
    assert bad.sum() == 0, bad.sum()
    assert np.abs(w - ow).max() <= 2e-5 * ow.max()
    assert np.abs(wn - own).max() <= 2e-5 * own.max()
    assert np.abs(gn - ogn).max() <= 1e-12 * ogn.max()
    assert np.abs(dn - odn).max() <= 2e-5 * odn.max()
    # without the nest's fields the engine must refuse rather than use the mother grid silently
    sc2 = {k: v for k, v in sc.items() if k not in ("lsprecn", "convprecn", "tccn", "ttn", "cloudsn", "cloudshn")}
    eng = Engine(sc2, compute_real_bytes=8, host_real_bytes=8)
    with pytest.raises(RuntimeError):
        eng.run(2)        # wetdepo runs from the second step on (timemanager.f90:164-169)
    eng.close()


def test_full_size_order_independence(built):
    """BASELINE config 3 at (half) its full size, where no CPU oracle finishes: 5e7 particles on the 361x181x138
    grid, Hanna + CBL + counter RNG.  Size-independent property: the result of a step depends on the particle
    (number, seed, step) only -- a run that re-sorts the particles by grid cell before every step and a run that
    never sorts give bit-identical states for every particle number (compared through order-sensitive checksums
    of the downloaded arrays, and exactly on a sample)."""
    from flexpart_amd.engine import Engine, RNG_PHILOX
    n = 50_000_000
    sc = syn.base_scenario(ctl=5.0, ifine=4, cblflag=1, nsteps=2)
    out = []
    for sort_interval in (0, 1):
        eng = Engine(sc, compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_PHILOX, seed=2024, max_particles=n,
                     sort_interval=sort_interval)
        eng.seed_particles(n, seed=77, frac_pbl=0.5)
        if sort_interval:
            eng.sort()
        eng.step_async(0)
        eng.step_async(int(sc["lsynctime"]))
        eng.sync()
        cnt = eng.counters()
        r = eng.download()
        eng.close()
        # a handful of particles per 1e8 end in the CBL scheme's own blow-up (cbl.f90 has no guard where its
        # bi-Gaussian weights leave [0,1]); they are terminated like particles that leave the domain
        assert cnt["n_bad_position"] == 0 and cnt["n_due"] >= 2 * n - 10 and cnt["n_left_domain"] <= 10
        alive = r["itra1"] != -999999999
        w = np.arange(1, 1001, dtype=np.float64)
        sums = {"alive": alive}
        for k in ("xtra1", "ytra1", "ztra1", "uzp"):
            a = np.where(alive, r[k], 0.0)
            assert np.all(np.isfinite(a))
            sums[k] = (float(a.sum()), float((a[: (n // 1000) * 1000].reshape(-1, 1000) * w).sum()), a[::100003].copy())
        sums["idt"] = int(r["idt"].astype(np.int64).sum())
        out.append(sums)
        del r
    a, b = out
    assert a["idt"] == b["idt"] and np.array_equal(a["alive"], b["alive"])
    for k in ("xtra1", "ytra1", "ztra1", "uzp"):
        assert a[k][0] == b[k][0] and a[k][1] == b[k][1], k
        assert np.array_equal(a[k][2], b[k][2]), k


def test_many_levels_and_three_species(built):
    """Upper end of the sizes: 200 model levels (the level search and the dynamic LDS layout of the Langevin kernel
    scale with nz) and three species of which two deposit and two decay -- the ragged species loops of the
    epilogue (timemanager.f90:642-696) and of the dry-deposition probability (advance.f90:582-599)."""
    sc = syn.small(n=3000, nx=40, ny=24, nz=200, nsteps=3, ctl=5.0, ifine=4, nspec=3)
    sc.update(lsettling=1, drydep=1, drydepspec=np.array([1, 0, 1], np.int32), density=np.array([2000.0, 0.0, 1500.0]),
              dquer=np.array([8.0, 0.0, 3.0]), vsetaver=np.array([-0.004, 0.0, -0.001]), cunningham=np.array([1.02, 1.0, 1.05]),
              decay=np.array([1.0e-6, 0.0, 2.0e-6]), xmass=np.array([1.0, 2.0, 0.5]))
    got, want = run_pair(sc, "r8")
    for g, w in zip(got, want):
        assert_close(g, w, 1e-9, 1e-7)
        scale = np.abs(w["xmass1"]).max()
        assert np.abs(g["xmass1"] - w["xmass1"]).max() <= 1e-12 * scale
    m0 = np.asarray(sc["xmass1"]).reshape(3, -1)
    assert (want[-1]["xmass1"][0] < m0[0]).any() and np.array_equal(want[-1]["xmass1"][1], m0[1])   # species 2 neither deposits nor decays


def test_edge_cases_empty_dead_and_not_due(built):
    """The loop's guards, timemanager.f90:531-535: no particles at all; particles that are not due at this
    itime (itra1 != itime) or terminated (itra1 = -999999999) are left bit-for-bit untouched; a single particle."""
    import ctypes as C
    from flexpart_amd.engine import Engine, RNG_TABLE_SEQ
    from flexpart_amd._lib import check
    from oracle.oracle import Oracle
    sc = syn.small(n=300, nx=40, ny=24, nz=30, nsteps=2, ctl=5.0, ifine=4)
    # 1) empty: numpart = 0
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_TABLE_SEQ)
    check(eng.lib.fpx_set_numpart(eng.h, C.c_int64(0)), "fpx_set_numpart")
    st = eng.step(0)
    assert st["n_due"] == 0 and st["n_initialized"] == 0
    eng.close()
    # 2) every third particle terminated, every third scheduled for a later time
    sc2 = dict(sc)
    itra1 = np.asarray(sc["itra1"]).copy()
    itra1[0::3] = -999999999
    itra1[1::3] = 900
    sc2["itra1"] = itra1
    eng = Engine(sc2, compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_TABLE_SEQ)
    before = eng.download()
    st = eng.step(0)
    after = eng.download()
    eng.close()
    assert st["n_due"] == int((itra1 == 0).sum())
    idle = itra1 != 0
    for k in ("xtra1", "ytra1", "ztra1", "uap", "ucp", "uzp", "us", "vs", "ws", "itra1", "idt", "cbt"):
        assert np.array_equal(before[k][idle], after[k][idle]), k
    assert np.all(after["itra1"][~idle] != 0)            # the due ones were advanced (or terminated)
    orc = Oracle(sc2, "r8")
    orc.lib.orc_set_parallel_semantics(orc.h, 1)
    want = orc.run(1)[-1]
    assert np.array_equal(after["itra1"], want["itra1"])
    assert np.abs(after["xtra1"] - want["xtra1"]).max() <= 1e-9 * np.abs(want["xtra1"]).max()
    # 3) one particle
    sc1 = syn.small(n=1, nx=40, ny=24, nz=30, nsteps=2, ctl=5.0, ifine=4)
    got, want = run_pair(sc1, "r8")
    for k in POS:
        assert abs(got[-1][k][0] - want[-1][k][0]) <= 1e-9 * max(abs(want[-1][k][0]), 1.0)


def test_release_during_the_run_after_a_locality_sort(built):
    """releaseparticles() adds particles while the model runs (timemanager.f90:246): the second half of the
    cloud is uploaded (fpx_upload_particles with first = numpart) after two steps and a locality sort of the
    first half.  The oracle carries all particles from the start, the late ones scheduled for the release time,
    which is what the reference's loop does with them (`if (itra1(j).ne.itime) cycle`, initialize() at itramem)."""
    from flexpart_amd.engine import Engine, RNG_TABLE_SEQ
    from oracle.oracle import Oracle
    n, half, t_rel = 3000, 1500, 1800
    sc = syn.small(n=n, nx=60, ny=40, nz=40, nsteps=4, ctl=5.0, ifine=4, seed=77)
    itra1 = np.asarray(sc["itra1"]).copy(); itramem = np.asarray(sc["itramem"]).copy()
    itra1[half:] = t_rel; itramem[half:] = t_rel
    sc.update(itra1=itra1, itramem=itramem)
    orc = Oracle(sc, "r8")
    orc.lib.orc_set_parallel_semantics(orc.h, 1)
    want = orc.run(4)[-1]

    def part(lo, hi):
        d = dict(sc)
        d["npart"] = hi - lo
        for k in ("xtra1", "ytra1", "ztra1", "itra1", "itramem", "npoint", "nclass", "idt", "uap", "ucp", "uzp", "us", "vs", "ws", "cbt"):
            if k in sc:
                d[k] = np.asarray(sc[k])[lo:hi]
        d["xmass1"] = np.asarray(sc["xmass1"]).reshape(-1, n)[:, lo:hi]
        return d
    eng = Engine(part(0, half), compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_TABLE_SEQ, max_particles=n)
    eng.step(); eng.step()
    eng.sort()
    eng.upload_particles_from_scenario(part(half, n), first=half)
    eng.step(); eng.step()
    got = eng.download()
    eng.close()
    assert got["xtra1"].size == n
    assert np.array_equal(got["itra1"], want["itra1"])
    for k in POS:
        assert np.abs(got[k] - want[k]).max() <= 1e-9 * np.abs(want[k]).max(), k


def test_long_run_stays_healthy(built):
    """80 synchronisation steps (counter RNG, CBL + Hanna, locality re-sorts, the wind window moved as getfields()
    would, more steps than the engine's internal timing-event pool holds): every particle stays finite, inside the
    domain and on the loop's schedule; nothing is flagged as a bad position."""
    from flexpart_amd.engine import Engine, RNG_PHILOX
    n = 20000
    sc = syn.small(n=n, nx=60, ny=40, nz=40, nsteps=80, ctl=5.0, ifine=4, cblflag=1)
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_PHILOX, sort_interval=5)
    lsync, window = int(sc["lsynctime"]), 10800
    for i in range(80):
        itime = i * lsync
        w0 = (itime // window) * window
        eng.set_windtime((w0, w0 + window), (1, 2))
        eng.step_async(itime)
    eng.sync()
    cnt = eng.counters()
    got = eng.download()
    eng.close()
    alive = got["itra1"] != -999999999
    assert cnt["n_bad_position"] == 0 and cnt["n_due"] >= 79 * alive.sum()
    assert alive.sum() >= 0.9 * n and np.all(got["itra1"][alive] == 80 * lsync)
    nx, ny = int(sc["grid"][0]), int(sc["grid"][1])
    for k in ("xtra1", "ytra1", "ztra1", "uap", "ucp", "uzp", "us", "vs", "ws"):
        assert np.all(np.isfinite(got[k][alive])), k
    assert got["xtra1"][alive].min() >= 0 and got["xtra1"][alive].max() <= nx - 1
    assert got["ytra1"][alive].min() >= 0 and got["ytra1"][alive].max() <= ny - 1
    assert got["ztra1"][alive].min() >= 0 and got["ztra1"][alive].max() <= float(np.asarray(sc["height"])[-1])
    # the cloud has spread: turbulence and wind did move the particles
    assert np.abs(got["xtra1"][alive] - np.asarray(sc["xtra1"])[alive]).max() > 1.0


def test_device_math_helpers_against_libm(built):
    """The 1-2 ulp fp64 helpers of the Langevin loop (fpx_device.hpp: m_expp, m_logp, m_sqrtp, m_rcp,
    m_rsqrt, m_cuberoot_parts) against numpy/libm.  Tolerance 4 ulp (8.9e-16 relative); for the
    logarithm the bound is absolute, 4e-16*max(1,|log x|), which is what x**y = exp(y*log x) needs."""
    import ctypes as C
    from flexpart_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(7)

    def probe(fn, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        rc = lib.fpx_math_probe(fn, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size)
        assert rc == 0
        return y

    ulp4 = 4 * 2.0 ** -52
    n = 200000
    x = np.concatenate([rng.uniform(-745.0, 700.0, n), rng.uniform(-40.0, 2.0, n), [0.0, -0.0, 1e-300, -1e-300, -800.0]])
    got, want = probe(0, x), np.exp(x)
    norm = want > 1e-300
    assert np.max(np.abs(got[norm] - want[norm]) / want[norm]) < ulp4
    assert np.all(np.abs(got[~norm] - want[~norm]) <= 1e-300)
    pos = np.concatenate([10.0 ** rng.uniform(-300, 300, n), rng.uniform(0.5, 2.0, n), 1.0 + rng.uniform(-1e-8, 1e-8, 1000), [1.0, 2.0, 0.5]])
    got, want = probe(1, pos), np.log(pos)
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 4e-16
    assert probe(1, np.array([0.0]))[0] == -np.inf and np.isnan(probe(1, np.array([-1.0]))[0])
    mid = np.concatenate([10.0 ** rng.uniform(-150, 150, n), rng.uniform(0.0, 4.0, n)])
    got, want = probe(2, np.concatenate([mid, [0.0]])), np.sqrt(np.concatenate([mid, [0.0]]))
    assert got[-1] == 0.0
    assert np.max(np.abs(got[:-1] - want[:-1]) / want[:-1]) < ulp4
    sgn = mid * rng.choice([-1.0, 1.0], mid.size)
    assert np.max(np.abs(probe(3, sgn) * sgn - 1.0)) < ulp4
    assert np.max(np.abs(probe(4, mid) * np.sqrt(mid) - 1.0)) < ulp4
    zz = np.concatenate([10.0 ** rng.uniform(-12, 0.3, n), [1.0]])
    assert np.max(np.abs(probe(8, zz) / zz ** 0.8 - 1.0)) < 3e-15     # zeta**0.8: two Newton steps, ~6 roundings
    edge = probe(8, np.array([0.0, 1e-300]))
    assert edge[0] == 0.0 and abs(edge[1] / 1e-240 - 1.0) < 1e-12      # outside the f32 seed range: exp(0.8*log x)
    # erf from a known exp(-x^2): absolute accuracy (the value only enters O(1) sums in cbl.f90:195-204)
    from scipy.special import erf as sp_erf
    xe = np.concatenate([rng.uniform(-7.0, 7.0, n), rng.uniform(-1e-3, 1e-3, 1000), [0.0, 6.5, -6.5, 30.0, -30.0]])
    assert np.max(np.abs(probe(7, xe) - sp_erf(xe))) < 1e-15
    # the two "cube roots" of cbl.f90:115-121 keep the reference's exponent 0.333333333 (not 1/3)
    sk = np.concatenate([10.0 ** rng.uniform(-30, 30, n), 10.0 ** rng.uniform(-60, 60, 1000)])
    lg = np.log(sk)
    tol = ulp4 * (1.0 + np.abs(lg))       # exp(y*log x) amplifies the rounding of log x by |y log x|
    assert np.max(np.abs(probe(5, sk) / np.exp(0.333333333 * lg) - 1.0) / tol) < 1.0
    assert np.max(np.abs(probe(6, sk) / np.exp(-2.0 * 0.333333333 * lg) - 1.0) / tol) < 1.0
    # the table-based helpers of the fine sub-step: exp with a 64-entry table of 2**(j/64) (4 ulp); the logarithm that is
    # only asked for an ABSOLUTE accuracy (1e-13: it feeds factors zeta**delta with |delta| ~ 1e-5); zeta**(-1/3) by Newton steps
    xt = np.concatenate([rng.uniform(-745.0, 700.0, n), rng.uniform(-40.0, 2.0, n), rng.uniform(-1e-3, 1e-3, 1000), [0.0, -0.0, 1e-300, -800.0]])
    got, want = probe(9, xt), np.exp(xt)
    norm = want > 1e-300
    assert np.max(np.abs(got[norm] - want[norm]) / want[norm]) < ulp4
    assert np.all(np.abs(got[~norm] - want[~norm]) <= 1e-300)
    pl = np.concatenate([10.0 ** rng.uniform(-37, 3, n), rng.uniform(0.0, 1.0, n)[1:], 1.0 + rng.uniform(-1e-8, 1e-8, 1000), [1.0, 0.5, 1e-37, 1e-3]])
    assert np.max(np.abs(probe(10, pl) - np.log(pl))) < 1e-13
    zc = np.concatenate([10.0 ** rng.uniform(-37, 2, n), rng.uniform(0.0, 1.0, n)[1:], [1.0, 1e-3]])
    assert np.max(np.abs(probe(11, zc) * np.cbrt(zc) - 1.0)) < ulp4


@pytest.mark.parametrize("rb,aerosol,waves", [(8, False, 3), (8, True, 3), (4, False, 4), (4, True, 4)])
def test_langevin_kernel_keeps_its_occupancy(built, rb, aerosol, waves):
    """The persistent grid of the Langevin kernel = resident blocks per CU x CUs: three blocks of four waves per CU in fp64, four in
    f32 -- also for the aerosol instances, whose stash (LDS) is the larger one: an LDS or register budget that silently costs a
    block per CU is a 15-30 % slower kernel ("LDS budget rule")."""
    from flexpart_amd.engine import Engine, RNG_PHILOX
    sc = syn.small(n=3000, nx=40, ny=24, nz=138, nsteps=1, ctl=5.0, ifine=4, cblflag=1)
    if aerosol:
        sc.update(lsettling=1, drydep=1, drydepspec=np.array([1], np.int32), density=np.array([2000.0]),
                  dquer=np.array([8.0]), vsetaver=np.array([-0.004]), cunningham=np.array([1.02]), xmass=np.array([1.0]))
    eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=RNG_PHILOX)
    eng.step()
    per_cu, grid = eng.info("pbl_blocks_per_cu"), eng.info("pbl_grid")
    eng.close()
    assert per_cu == waves and grid % waves == 0 and grid >= waves * 64, (per_cu, grid)


def test_options_and_info_are_checked(built):
    """fpx_set_option / fpx_get_info: unknown names and malformed values are refused (FPX_ERR_ARG), nothing is read from the
    environment, and a knob that is set is what the engine reports."""
    import os
    from flexpart_amd.engine import Engine, RNG_PHILOX
    sc = syn.small(n=500, nx=20, ny=12, nz=10, nsteps=1, ctl=5.0, ifine=4)
    os.environ["FPX_BLEND_MIN"] = "1"          # round 3's switch: must be ignored now
    try:
        eng = Engine(sc, rng_mode=RNG_PHILOX)
        assert eng.info("time_blended_packs") == 0
        for name, value in (("no_such_option", "1"), ("pbl_slices", "4,x"), ("pbl_slices", "-3"), ("permute", "sideways"), ("pbl_drain_lanes", "65"),
                            ("pbl_cost_buckets", "9"), ("verbose", "yes")):
            with pytest.raises(Exception, match="unknown option or malformed value"):
                eng.set_option(name, value)
        with pytest.raises(Exception, match="unknown name"):
            eng.info("no_such_info")
        eng.set_option("pbl_slices", "5,5,0")
        assert eng.info("pbl_launches_per_step") == 3
        eng.set_option("pbl_slices", "0")
        assert eng.info("pbl_launches_per_step") == 1
        eng.step()
        assert eng.info("blended_steps") == 0 and eng.info("pbl_blocks_per_cu") == 3
        eng.close()
        for bad in (dict(blend_mode=3), dict(global_particles=-1), dict(pbl_slice_passes=-2)):
            with pytest.raises(Exception):
                Engine(sc, rng_mode=RNG_PHILOX, **bad)
    finally:
        del os.environ["FPX_BLEND_MIN"]
