"""Deterministic synthetic scenarios for the particle-advection path.

Everything here is built from integer arithmetic plus IEEE-exact float64
operations (+ - * / sqrt, comparisons) only -- no libm transcendentals -- so a
scenario regenerates bit-identically on any box (the GPU box has no access to
the reference tree or to scenario files made elsewhere).

A scenario is a plain dict whose keys follow the reference's own variable names
(com_mod.f90 / par_mod.f90 of the reference); see `Scenario keys` below.  Field
arrays are float64, C-ordered with shape (2, nz, ny, nx) (time slot, level, y,
x), i.e. the reference's column-major (x, y, z, slot) order with only the used
nx*ny*nz extent kept.

The shapes follow SURVEY.md section 8(d): ECMWF-like 361x181x138 global grid,
two wind-field time slots 10800 s apart, lsynctime 900 s.
"""
from __future__ import annotations

import numpy as np

# constants of the reference (par_mod.f90:59-60,76-79,112,123,213,254)
PI_REF = 3.14159265
R_EARTH = 6.371e6
HMIXMIN, HMIXMAX = 100.0, 4500.0
SWITCHNORTH, SWITCHSOUTH = 75.0, -75.0
MAXRAND = 1000000
MINSTEP = 1
DEAD = -999999999


def derive_switches(ctl_cmd: float, ifine_cmd: int, cblflag: int, lsynctime: int):
    """COMMAND-file values -> the run-time switches the hot path reads.

    Mirrors readcommand.f90:244-272,379-385 of the reference: returns a dict
    with method, mintime, turbswitch, ifine, ctl (already inverted), lsynctime.
    """
    ifine = max(int(ifine_cmd), 1)
    ctl = float(ctl_cmd)
    if cblflag == 1:
        turbswitch = True
        lsynctime = min(lsynctime, 1200)           # maxtl, com_mod.f90:752
        if ctl < 5:
            ctl = 5.0
        if ifine * ctl < 50:
            ifine = int(50.0 / ctl) + 1
    else:
        if ctl >= 0.1:
            turbswitch = True
        else:
            turbswitch = False
            ifine = 1
    ctl_inv = 1.0 / ctl
    if ctl_inv > 0.0:
        method, mintime = 1, MINSTEP
    else:
        method, mintime = 0, lsynctime
    return dict(method=method, mintime=mintime, turbswitch=int(turbswitch),
                ifine=ifine, ctl=ctl_inv, lsynctime=lsynctime, cblflag=int(cblflag))


# --------------------------------------------------------------------------
# exact building blocks
# --------------------------------------------------------------------------
def _wave(num, den):
    """C1 'parabola sine' of phase num/den (integer arrays), range [-1, 1]."""
    t = (np.asarray(num, dtype=np.int64) % int(den)).astype(np.float64) / float(den)
    return np.where(t < 0.5, 16.0 * t * (0.5 - t), -16.0 * (t - 0.5) * (1.0 - t))


def _splitmix64(n, seed):
    """n uint64 values from SplitMix64 counters (vectorised, wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64(seed))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform01(n, seed):
    """float64 in [0, 1) with 53 random bits, exact on every platform."""
    return (_splitmix64(n, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def make_height(nz, top=80000.0, lin=0.05):
    """Stretched model-level heights [m], height[0] = 0 (verttransform output shape)."""
    s = np.arange(nz, dtype=np.float64) / float(nz - 1)
    return top * (lin * s + (1.0 - lin) * s * s * s)


def nmixz_from_height(height):
    """First level (1-based) above hmixmax, as verttransform_ecmwf.f90:186-192."""
    for kz in range(len(height)):
        if height[kz] > HMIXMAX:
            return kz + 1
    return len(height)


# --------------------------------------------------------------------------
# fields
# --------------------------------------------------------------------------
def make_fields(nx, ny, nz, height, *, uniform=None, polar=False, nspec=1, hmix_const=None):
    """Smooth synthetic met fields on an (nx, ny, nz) grid with 2 time slots.

    uniform: None for the ECMWF-shaped analytic fields, or a dict
    {u, v, w, rho, hmix, ustar, wstar, oli} for spatially uniform fields (the
    reference's own validation scenario, mpi_mod.f90:2903-2975).
    """
    f = {}
    shp3 = (2, nz, ny, nx)
    shp2 = (2, ny, nx)
    if uniform is not None:
        f["uu"] = np.full(shp3, float(uniform.get("u", 10.0)))
        f["vv"] = np.full(shp3, float(uniform.get("v", 0.0)))
        f["ww"] = np.full(shp3, float(uniform.get("w", 0.0)))
        f["rho"] = np.full(shp3, float(uniform.get("rho", 1.3)))
        f["drhodz"] = np.full(shp3, float(uniform.get("drhodz", 0.0)))
        f["tt"] = np.full(shp3, float(uniform.get("tt", 273.0)))
        f["hmix"] = np.full(shp2, float(uniform.get("hmix", 10000.0)))
        f["ustar"] = np.full(shp2, float(uniform.get("ustar", 1.0)))
        f["wstar"] = np.full(shp2, float(uniform.get("wstar", 1.0)))
        f["oli"] = np.full(shp2, float(uniform.get("oli", 0.01)))
        f["tropopause"] = np.full(shp2, float(uniform.get("tropopause", 10000.0)))
        f["vdep"] = np.full((2, nspec, ny, nx), float(uniform.get("vdep", 0.002)))
        if polar:
            f["uupol"] = f["uu"].copy()
            f["vvpol"] = f["vv"].copy()
        return f

    per = nx - 1                                   # cyclic period in x (column nx-1 == column 0)
    i = np.arange(nx, dtype=np.int64)[None, None, :]
    j = np.arange(ny, dtype=np.int64)[None, :, None]
    k = np.arange(nz, dtype=np.int64)[:, None, None]
    s = np.arange(ny, dtype=np.float64)[None, :, None] / float(ny - 1)
    clat = 4.0 * s * (1.0 - s)                     # 0 at the poles, 1 at the equator
    z = np.asarray(height, dtype=np.float64)[:, None, None]
    zf = z / float(height[-1])
    a = z / 8000.0
    den = 1.0 + a + 0.5 * a * a
    f["uu"] = np.empty(shp3); f["vv"] = np.empty(shp3); f["ww"] = np.empty(shp3)
    f["rho"] = np.empty(shp3); f["drhodz"] = np.empty(shp3); f["tt"] = np.empty(shp3)
    for m in range(2):
        sh = 10 * m                                # second slot: pattern shifted by 10 columns
        f["uu"][m] = (20.0 * clat * (1.0 + 0.3 * _wave(k, nz))
                      + 5.0 * _wave(3 * (i + sh), per) + 0.0 * j)
        f["vv"][m] = 5.0 * _wave(2 * (i + sh) + 0 * j, per) * clat + 1.5 * _wave(k + 3 * j, 40)
        f["ww"][m] = (0.05 * _wave(i + sh, per) * _wave(2 * j + 0 * i, ny - 1)
                      * 4.0 * zf * (1.0 - zf))
        r = (1.2 + 0.02 * m) / den + 0.0 * (i + j)
        f["rho"][m] = r * (1.0 + 0.01 * _wave(i + 2 * j, per))
        f["drhodz"][m] = -(1.2 + 0.02 * m) * (1.0 + a) / 8000.0 / (den * den) + 0.0 * (i + j)
        f["tt"][m] = np.maximum(288.0 - 0.0065 * z + 2.0 * _wave(i + sh + j, per), 216.0)
    i2 = i[0]; j2 = j[0]; c2 = clat[0]
    f["hmix"] = np.empty(shp2); f["ustar"] = np.empty(shp2); f["wstar"] = np.empty(shp2)
    f["oli"] = np.empty(shp2); f["tropopause"] = np.empty(shp2)
    f["vdep"] = np.empty((2, nspec, ny, nx))
    for m in range(2):
        sh = 10 * m
        if hmix_const is not None:
            f["hmix"][m] = float(hmix_const)
        else:
            hm = 300.0 + 1200.0 * (0.5 + 0.5 * _wave(i2 + sh, per) * c2) + 300.0 * _wave(3 * j2 + i2, 64)
            f["hmix"][m] = np.minimum(np.maximum(hm, HMIXMIN), HMIXMAX)
        f["ustar"][m] = 0.1 + 0.4 * c2 + 0.05 * _wave(5 * (i2 + sh), per)
        olw = _wave(2 * (i2 + sh) + j2, per)
        f["oli"][m] = 0.02 * olw * (0.25 + 0.75 * c2)
        # convective velocity scale: sizeable where the surface layer is unstable (1/L < 0),
        # small but non-zero elsewhere (the CBL scheme divides by powers of w*)
        f["wstar"][m] = 0.1 + 1.9 * np.maximum(0.0, -olw) + 0.0 * c2
        f["tropopause"][m] = 9000.0 + 7000.0 * c2 + 0.0 * i2
        for ks in range(nspec):
            f["vdep"][m, ks] = 0.002 * (ks + 1) + 0.001 * _wave(i2 + sh + j2, per)
    if xcyclic_ok(nx):
        for key in ("uu", "vv", "ww", "rho", "drhodz", "tt", "hmix", "ustar", "wstar", "oli",
                    "tropopause", "vdep"):
            f[key][..., nx - 1] = f[key][..., 0]   # cyclic duplicate column (gridcheck: nx = nxfield+1)
    if polar:
        # any smooth field serves parity: the real ones come from verttransform (out of scope)
        f["uupol"] = 0.7 * f["uu"] + 0.2 * f["vv"]
        f["vvpol"] = 0.7 * f["vv"] - 0.2 * f["uu"]
    return f


def xcyclic_ok(nx):
    return nx >= 3


# --------------------------------------------------------------------------
# particles
# --------------------------------------------------------------------------
def make_particles(n, nx, ny, height, hmix, *, seed=0x5EED, lat_margin_cells=None,
                   frac_pbl=0.5, zmax=12000.0, nspec=1, itime0=0):
    """Fixed-seed particle cloud over the whole grid.

    Half the particles (frac_pbl) are placed below the local mixing height so
    that both branches of the trajectory step are exercised.
    """
    ux = _uniform01(n, seed + 1)
    uy = _uniform01(n, seed + 2)
    uz = _uniform01(n, seed + 3)
    us = _uniform01(n, seed + 4)
    eps = 361.0 / 3.0e5
    if lat_margin_cells is None:
        lat_margin_cells = 0.03 * (ny - 1)
    x = eps + ux * (float(nx - 1) - 2.0 * eps)
    y = lat_margin_cells + uy * (float(ny - 1) - 2.0 * lat_margin_cells)
    ix = np.minimum(x.astype(np.int64), nx - 2)
    jy = np.minimum(y.astype(np.int64), ny - 2)
    hloc = np.maximum.reduce([hmix[0, jy, ix], hmix[0, jy, ix + 1], hmix[0, jy + 1, ix],
                              hmix[0, jy + 1, ix + 1], hmix[1, jy, ix], hmix[1, jy, ix + 1],
                              hmix[1, jy + 1, ix], hmix[1, jy + 1, ix + 1]])
    in_pbl = us < frac_pbl
    z = np.where(in_pbl, 10.0 + uz * np.maximum(hloc - 20.0, 1.0),
                 hloc + 10.0 + uz * (zmax - hloc - 10.0))
    p = dict(npart=n, xtra1=x, ytra1=y, ztra1=z,
             itra1=np.full(n, itime0, np.int32), itramem=np.full(n, itime0, np.int32),
             npoint=np.ones(n, np.int32), nclass=np.ones(n, np.int32),
             xmass1=np.ones((nspec, n), np.float64))
    return p


def point_release(n, xlon, ylat, z, dx, dy, xlon0, ylat0, *, nspec=1, itime0=0):
    """All particles at one point (options/RELEASES default case of the reference)."""
    x = np.full(n, (xlon - xlon0) / dx)
    y = np.full(n, (ylat - ylat0) / dy)
    return dict(npart=n, xtra1=x, ytra1=y, ztra1=np.full(n, float(z)),
                itra1=np.full(n, itime0, np.int32), itramem=np.full(n, itime0, np.int32),
                npoint=np.ones(n, np.int32), nclass=np.ones(n, np.int32),
                xmass1=np.full((nspec, n), 1.0 / n))


# --------------------------------------------------------------------------
# scenarios
# --------------------------------------------------------------------------
def base_scenario(nx=361, ny=181, nz=138, *, global_grid=True, polar=False, nspec=1,
                  ctl=5.0, ifine=4, cblflag=0, lsynctime=900, uniform=None, hmix_const=None,
                  turb_off=False, nsteps=2, ldirect=1):
    """Grid + switches + fields; particles are added by the caller."""
    height = make_height(nz, top=80000.0 if nz >= 60 else 30000.0,
                         lin=0.05 if nz >= 60 else 0.15)
    dx = 360.0 / (nx - 1) if global_grid else 1.0
    dy = 180.0 / (ny - 1) if global_grid else 1.0
    xlon0 = -180.0 if global_grid else -20.0
    ylat0 = -90.0 if global_grid else 20.0
    sw = derive_switches(ctl, ifine, cblflag, lsynctime)
    sc = dict(
        grid=np.array([nx, ny, nz], np.int32),
        geom=np.array([dx, dy, xlon0, ylat0], np.float64),
        globalflags=np.array([int(global_grid), int(global_grid and polar),
                              int(global_grid and polar)], np.int32),
        height=height, nmixz=nmixz_from_height(height),
        # backward runs: lsynctime and the wind window run towards negative times (readcommand.f90:631,
        # getfields.f90 orders wftime by ldirect); mintime stays positive (readcommand.f90:381-384 come first)
        memtime=np.array([0, 10800 * ldirect], np.int32), memind=np.array([1, 2], np.int32),
        ldirect=ldirect, lsynctime=sw["lsynctime"] * ldirect, method=sw["method"], mintime=sw["mintime"],
        ctl=sw["ctl"], ifine=sw["ifine"], turbswitch=sw["turbswitch"], cblflag=sw["cblflag"],
        mdomainfill=0, lsettling=0, nspec=nspec,
        drydep=0, drydepspec=np.zeros(nspec, np.int32),
        density=np.zeros(nspec), dquer=np.zeros(nspec), vsetaver=np.zeros(nspec),
        cunningham=np.ones(nspec), decay=np.zeros(nspec),
        turbpar=np.array([0.0, 0.0, 0.0]) if turb_off else np.array([50.0, 0.1, 0.16]),
        lage=np.array([999999999], np.int32),
        nsteps=nsteps, itime0=0,
    )
    sc.update(make_fields(nx, ny, nz, height, uniform=uniform, polar=polar, nspec=nspec,
                          hmix_const=hmix_const))
    return sc


def config1(n=10000, nsteps=2):
    """BASELINE config 1: 10k particles, point release, uniform wind, no turbulence.

    Uniform values of the reference's own validation set (mpi_mod.f90:2940-2973: u=10, v=0,
    w=0, rho=1.3); 'no turbulence' is obtained the reference's run-time way (SURVEY appendix
    A): particles above the mixing height and d_trop = d_strat = turbmesoscale = 0.
    """
    sc = base_scenario(uniform=dict(u=10.0, v=0.0, w=0.0, rho=1.3, hmix=HMIXMIN, ustar=1.0,
                                    wstar=1.0, oli=0.01, tropopause=10000.0),
                       ctl=-5.0, ifine=4, turb_off=True, nsteps=nsteps)
    g = sc["geom"]
    sc.update(point_release(n, 0.0, 20.0, 150.0, g[0], g[1], g[2], g[3]))
    return sc


def config2(n=10_000_000, nsteps=2, seed=0x5EED, nx=361, ny=181, nz=138):
    """BASELINE config 2: advance + interpol_wind only (all particles above the PBL)."""
    sc = base_scenario(nx, ny, nz, ctl=-5.0, ifine=4, turb_off=True, hmix_const=HMIXMIN,
                       nsteps=nsteps)
    sc.update(make_particles(n, nx, ny, sc["height"], sc["hmix"], seed=seed, frac_pbl=0.0))
    return sc


def config3(n=100_000_000, nsteps=2, seed=0x5EED, nx=361, ny=181, nz=138, cblflag=1,
            ctl=5.0, ifine=4):
    """BASELINE config 3: full Hanna turbulence + CBL, PBL sub-stepping."""
    sc = base_scenario(nx, ny, nz, ctl=ctl, ifine=ifine, cblflag=cblflag, nsteps=nsteps)
    sc.update(make_particles(n, nx, ny, sc["height"], sc["hmix"], seed=seed, frac_pbl=0.5))
    return sc


def small(n=2000, nx=40, ny=24, nz=30, **kw):
    """A small grid + cloud for fast parity tests (same generators, reduced sizes)."""
    cbl = kw.pop("cblflag", 0)
    ctl = kw.pop("ctl", 5.0)
    ifine = kw.pop("ifine", 4)
    nsteps = kw.pop("nsteps", 3)
    seed = kw.pop("seed", 1234)
    frac_pbl = kw.pop("frac_pbl", 0.5)
    zmax = kw.pop("zmax", 12000.0)
    lat_margin = kw.pop("lat_margin_cells", None)
    sc = base_scenario(nx, ny, nz, ctl=ctl, ifine=ifine, cblflag=cbl, nsteps=nsteps, **kw)
    sc.update(make_particles(n, nx, ny, sc["height"], sc["hmix"], seed=seed, frac_pbl=frac_pbl,
                             zmax=zmax, nspec=int(sc["nspec"]), lat_margin_cells=lat_margin))
    return sc


def add_outgrid(sc, nxg=36, nyg=18, nzg=5, *, outlon0=None, outlat0=None, dxout=None, dyout=None,
                ind_samp=-1, old_fraction=0.5, age=20000):
    """Output grid of the concentration sampling (OUTGRID namelist shape, readoutgrid.f90) plus
    a share of particles released `age` seconds ago, so that both the direct-cell and the
    4-cell-kernel branch of conccalc (conccalc.f90:171) are exercised."""
    nx, ny, _ = (int(v) for v in sc["grid"])
    dx, dy, xlon0, ylat0 = (float(v) for v in sc["geom"])
    if outlon0 is None:
        outlon0 = xlon0 + 0.05 * (nx - 1) * dx
    if outlat0 is None:
        outlat0 = ylat0 + 0.10 * (ny - 1) * dy
    if dxout is None:
        dxout = 0.9 * (nx - 1) * dx / nxg
    if dyout is None:
        dyout = 0.8 * (ny - 1) * dy / nyg
    sc["outgrid"] = np.array([nxg, nyg, nzg], np.int32)
    sc["outgeom"] = np.array([dxout, dyout, outlon0, outlat0], np.float64)
    sc["outheight"] = np.array([100.0, 500.0, 1500.0, 5000.0, 20000.0][:nzg] if nzg <= 5
                               else 100.0 * (np.arange(nzg) + 1.0) ** 2)
    sc["concflags"] = np.array([ind_samp, 0], np.int32)
    sc["outtimes"] = np.array([3600, 3600], np.int32)
    n = int(sc["npart"])
    old = _uniform01(n, 0xA6E) < old_fraction
    itramem = np.asarray(sc["itramem"]).copy()
    itramem[old] = int(sc["itime0"]) - int(age)
    sc["itramem"] = itramem.astype(np.int32)
    return sc


def add_release_points(sc, xmass, npart_rel, *, ioutputforeachrelease=1, lage=None, max_age=None,
                       nclassunc=1, tiny_every=0, tiny_factor=5.0e-5, near_every=0, near_factor=1.03e-4):
    """Several release points (RELEASES with more than one &RELEASE block): point_mod xmass(numpoint,nspec)
    (given as [nspec][numpoint]) and npart(numpoint); every particle gets a release point npoint(j), the mass
    xmass(kp,ks)/npart(kp) of its point (readreleases.f90 / releaseparticles.f90:148-155), an uncertainty class
    nclass(j) in 1..nclassunc and -- with max_age -- a release time up to max_age seconds before the start, so
    that particles fall into several age classes and some exceed lage(nageclass) (timemanager.f90:701-707).
    tiny_every / near_every: every k-th particle carries tiny_factor / near_factor of its nominal mass, i.e. lies
    below / just above minmass = 1e-4 (par_mod.f90:213) of it: terminated at once / after some decay
    (timemanager.f90:681-686)."""
    nspec = int(sc["nspec"])
    xm = np.asarray(xmass, dtype=np.float64).reshape(nspec, -1)
    numpoint = xm.shape[1]
    npt = np.asarray(npart_rel, dtype=np.int32).ravel()
    assert npt.size == numpoint
    n = int(sc["npart"])
    h = _splitmix64(n, 0xBEEF)
    kp = (h % np.uint64(numpoint)).astype(np.int64)                  # 0-based release point
    sc["numpoint"] = numpoint
    sc["xmass"] = xm
    sc["npart_rel"] = npt
    sc["npoint"] = (kp + 1).astype(np.int32)
    m = xm[:, kp] / npt[kp].astype(np.float64)[None, :]
    idx = np.arange(n)
    if tiny_every:
        m[:, idx % tiny_every == 1] *= tiny_factor
    if near_every:
        m[:, idx % near_every == 2] *= near_factor
    sc["xmass1"] = m
    sc["nclassunc"] = int(nclassunc)
    sc["nclass"] = (1 + ((h >> np.uint64(20)) % np.uint64(nclassunc)).astype(np.int64)).astype(np.int32)
    if lage is not None:
        sc["lage"] = np.asarray(lage, np.int32)
    if max_age:
        age = ((h >> np.uint64(32)) % np.uint64(int(max_age))).astype(np.int64)
        sc["itramem"] = (int(sc["itime0"]) - age * int(sc["ldirect"])).astype(np.int32)
    if "concflags" in sc:
        sc["concflags"] = np.array([int(sc["concflags"][0]), int(ioutputforeachrelease)], np.int32)
    return sc


def add_outgrid_nest(sc, nxn=30, nyn=20):
    """Nested output grid (OUTGRID_NEST, readoutgrid_nest.f90): finer cells over the middle of the
    mother output grid, so that particles fall inside, on its border and outside of it."""
    nxg, nyg, _ = (int(v) for v in sc["outgrid"])
    dxout, dyout, outlon0, outlat0 = (float(v) for v in sc["outgeom"])
    sc["outgridn"] = np.array([nxn, nyn], np.int32)
    sc["outgeomn"] = np.array([0.4 * nxg * dxout / nxn, 0.5 * nyg * dyout / nyn,
                               outlon0 + 0.3 * nxg * dxout, outlat0 + 0.25 * nyg * dyout], np.float64)
    return sc


def add_receptors(sc, m=4):
    """Receptor points (RECEPTORS, readreceptors.f90:88-92): positions in grid coordinates and the
    area of a dx*dy cell there; placed on particles so that the parabolic kernel finds some."""
    x = np.asarray(sc["xtra1"], dtype=np.float64)
    y = np.asarray(sc["ytra1"], dtype=np.float64)
    dx, dy, xlon0, ylat0 = (float(v) for v in sc["geom"])
    idx = (np.arange(m, dtype=np.int64) * 7919 + 11) % x.size
    xr = np.float32(x[idx]).astype(np.float64)
    yr = np.float32(y[idx]).astype(np.float64)
    r_earth, pi180 = 6.371e6, 3.14159265 / 180.0
    ylat = ylat0 + yr * dy
    area = (r_earth * np.cos(ylat * pi180) * dx * pi180) * (r_earth * dy * pi180)
    sc["receptors"] = np.concatenate([xr, yr, np.float32(area).astype(np.float64)])
    return sc


def _wet_fields(nx, ny, nz, z, phase):
    """lsprec, convprec, tcc [2][ny][nx]; clouds [2][nz][ny][nx]; cloudsh [2][ny][nx] (exact arithmetic)."""
    per = nx - 1
    i = np.arange(nx, dtype=np.int64)[None, :]
    j = np.arange(ny, dtype=np.int64)[:, None]
    lsprec = np.empty((2, ny, nx)); convprec = np.empty((2, ny, nx)); tcc = np.empty((2, ny, nx))
    clouds = np.zeros((2, nz, ny, nx), np.int32)
    cloudsh = np.zeros((2, ny, nx), np.int32)
    for m in range(2):
        sh = 7 * m + phase
        lsprec[m] = 12.0 * np.maximum(0.0, _wave(2 * (i + sh) + j, per)) ** 2
        convprec[m] = 6.0 * np.maximum(0.0, _wave(3 * (i + sh) + 2 * j, per))
        tcc[m] = 0.15 + 0.8 * np.abs(_wave(i + sh + 3 * j, per))
        raining = (lsprec[m] + convprec[m]) > 0.01
        for k in range(nz):
            cls = 6 if z[k] < 1200.0 else (3 if z[k] < 5000.0 else 0)
            clouds[m, k] = np.where(raining, cls, 0)
        cloudsh[m] = np.where(raining, 3800, 0)
    return lsprec, convprec, tcc, clouds, cloudsh


def add_wet_nest(sc):
    """The nest's own precipitation, cloud and temperature fields (lsprecn, convprecn, tccn, cloudsn,
    cloudshn, ttn; get_wetscav.f90:126-128,150-151,197-199); after add_nest and add_wet."""
    nxn, nyn = (int(v) for v in sc["nest"])
    nz = int(sc["grid"][2])
    lsp, cvp, tcc, cl, clh = _wet_fields(nxn, nyn, nz, np.asarray(sc["height"]), 3)
    f = make_fields(nxn, nyn, nz, sc["height"], nspec=int(sc["nspec"]))
    sc.update(lsprecn=lsp, convprecn=cvp, tccn=tcc, cloudsn=cl, cloudshn=clh, ttn=f["tt"] - 4.0)
    return sc


def add_wet(sc, *, gas=False):
    """Precipitation / cloud fields and wet-scavenging species parameters (readspecies.f90 names).

    clouds (int8 per level, verttransform's cloud classification): >= 4 below a precipitating
    cloud, 2-3 inside it, 0-1 none (get_wetscav.f90:150,206,251).  gas=False: an aerosol
    (dquer > 0, crain/csnow below cloud, ccn/in inside); gas=True: a soluble gas (weta/wetb,
    henry)."""
    nx, ny, nz = (int(v) for v in sc["grid"])
    lsprec, convprec, tcc, clouds, cloudsh = _wet_fields(nx, ny, nz, np.asarray(sc["height"]), 0)
    sc.update(lsprec=lsprec, convprec=convprec, tcc=tcc, clouds=clouds, cloudsh=cloudsh, wetdep=1,
              wetdepspec=np.array([1], np.int32))
    if gas:
        sc.update(weta_gas=np.array([2.0e-5]), wetb_gas=np.array([0.62]), crain_aero=np.array([0.0]),
                  csnow_aero=np.array([0.0]), ccn_aero=np.array([0.0]), in_aero=np.array([0.0]),
                  henry=np.array([1.0e5]))
    else:
        sc.update(weta_gas=np.array([0.0]), wetb_gas=np.array([0.0]), crain_aero=np.array([1.0]),
                  csnow_aero=np.array([1.0]), ccn_aero=np.array([0.9]), in_aero=np.array([0.1]),
                  henry=np.array([0.0]))
        if float(np.asarray(sc["dquer"])[0]) <= 0.0:
            sc["dquer"] = np.array([8.0])
    return sc


def add_nest(sc, ix0=None, jy0=None, ix1=None, jy1=None, factor=2):
    """One nested grid of `factor` times the mother resolution over mother cells [ix0,ix1]x[jy0,jy1]
    (com_mod.f90:464-541 arrays uun, vvn, ... ; geometry as gridcheck_nests.f90 derives it)."""
    nx, ny, nz = (int(v) for v in sc["grid"])
    dx, dy, xlon0, ylat0 = (float(v) for v in sc["geom"])
    ix0 = nx // 4 if ix0 is None else ix0
    ix1 = (2 * nx) // 3 if ix1 is None else ix1
    jy0 = ny // 4 if jy0 is None else jy0
    jy1 = (3 * ny) // 4 if jy1 is None else jy1
    nxn = factor * (ix1 - ix0) + 1
    nyn = factor * (jy1 - jy0) + 1
    f = make_fields(nxn, nyn, nz, sc["height"], nspec=int(sc["nspec"]))
    sc["nest"] = np.array([nxn, nyn], np.int32)
    sc["nestgeom"] = np.array([dx / factor, dy / factor, xlon0 + ix0 * dx, ylat0 + jy0 * dy], np.float64)
    for k, kn in (("uu", "uun"), ("vv", "vvn"), ("ww", "wwn"), ("rho", "rhon"), ("drhodz", "drhodzn"),
                  ("hmix", "hmixn"), ("ustar", "ustarn"), ("wstar", "wstarn"), ("oli", "olin"),
                  ("tropopause", "tropopausen"), ("vdep", "vdepn")):
        sc[kn] = f[k] * (1.1 if k in ("uu", "vv") else 1.0)   # not the mother's values: a wrong grid choice shows
    sc["par_nxmax"] = 721        # the reference variant with nests is built from par_mod_meteoswiss.f90
    return sc


# --------------------------------------------------------------------------
# model-level input of verttransform_ecmwf
# --------------------------------------------------------------------------
def model_levels(nx=361, ny=181, nz=138, *, global_grid=True, polar=False, phase=0):
    """ECMWF-shaped hybrid-level input as readwind_ecmwf leaves it: level 1 = surface,
    pressure of level k = akz[k] + bkz[k]*ps, nuvz = nwz = nz (gridcheck_ecmwf.f90 adds the
    surface level and sets nz = nuvz).  Arrays are compact [nz][ny][nx]; `phase` shifts the
    pattern (a second time slot).  Mountains (low ps) give columns whose top lies below the
    reference column's, which exercises the copy-the-top-level branch of the transform."""
    per = nx - 1
    i = np.arange(nx, dtype=np.int64)[None, None, :] + int(phase)
    j = np.arange(ny, dtype=np.int64)[None, :, None]
    k = np.arange(nz, dtype=np.int64)[:, None, None]
    s = np.arange(ny, dtype=np.float64)[None, :, None] / float(ny - 1)
    clat = 4.0 * s * (1.0 - s)
    # hybrid coefficients: eta from 1 (surface) to exp(-7) (about 90 Pa), layers thinnest at the ground
    eta = np.exp(-7.0 * (np.arange(nz, dtype=np.float64) / float(nz - 1)) ** 1.6)
    bk = eta ** 1.5
    bk[0] = 1.0
    ak = 101325.0 * (eta - bk)
    ak[0] = 0.0
    # surface pressure: a few 'mountain ranges' down to about 620 hPa, most columns above 1000 hPa
    mount = np.maximum(0.0, _wave(2 * i[0] + 3 * j[0], per)) * np.maximum(0.0, _wave(3 * j[0] + 7, 2 * (ny - 1)))
    ps = 101500.0 + 800.0 * _wave(i[0] + 2 * j[0], per) - 39000.0 * mount * mount
    lnp = np.log((ak[:, None, None] + bk[:, None, None] * ps[None]) / 101325.0)       # <= ~0
    zapprox = -7400.0 * lnp
    tth = np.maximum(288.0 - 0.0065 * zapprox, 216.0) + 3.0 * _wave(i + j + 2 * k, per) + 10.0 * (clat - 0.5)
    qvh = 0.012 * np.exp(-zapprox / 2500.0) * (0.6 + 0.4 * _wave(2 * i + j, per))
    uuh = 25.0 * clat * (1.0 + 0.3 * _wave(k, nz)) + 6.0 * _wave(3 * i, per) + 0.0 * j
    vvh = 6.0 * _wave(2 * i + 0 * j, per) * clat + 2.0 * _wave(k + 3 * j, 40)
    wwh = 0.4 * _wave(i, per) * _wave(2 * j + 0 * i, ny - 1) * np.exp(lnp)            # eta-dot * dp/deta, Pa/s
    pvh = 1.0e-6 * (1.0 + 0.5 * _wave(i + k, per)) * np.exp(zapprox / 12000.0) * (2.0 * s - 1.0)
    tt2 = tth[0] + 0.5 * _wave(3 * i[0] + j[0], per)
    td2 = tt2 - 3.0 - 2.0 * (1.0 + _wave(i[0] + 5 * j[0], per))
    out = dict(akz=ak, bkz=bk, aknew=ak.copy(), bknew=bk.copy(), ps=ps, tt2=tt2, td2=td2,
               tth=tth, qvh=qvh, uuh=uuh, vvh=vvh, pvh=pvh, wwh=wwh)
    if xcyclic_ok(nx) and global_grid:
        for key in ("ps", "tt2", "td2", "tth", "qvh", "uuh", "vvh", "pvh", "wwh"):
            out[key][..., nx - 1] = out[key][..., 0]
    dx = 360.0 / (nx - 1) if global_grid else 1.0
    dy = 180.0 / (ny - 1) if global_grid else 1.0
    out["grid"] = np.array([nx, ny, nz], np.int32)
    out["geom"] = np.array([dx, dy, -180.0 if global_grid else -20.0, -90.0 if global_grid else 20.0], np.float64)
    out["globalflags"] = np.array([int(global_grid), int(global_grid and polar), int(global_grid and polar)], np.int32)
    return out


# --------------------------------------------------------------------------
# particle dump (partoutput.f90): the extra fields it interpolates
# --------------------------------------------------------------------------
def add_partoutput_fields(sc, itime=None, dead_every=7):
    """oro, pv, qv for a scenario that already has height, rho, tt, hmix, tropopause and particles; every
    `dead_every`-th particle is made not due (terminated) so that the dump has to skip it."""
    nx, ny, nz = (int(v) for v in sc["grid"])
    per = nx - 1
    i = np.arange(nx, dtype=np.int64)[None, None, :]
    j = np.arange(ny, dtype=np.int64)[None, :, None]
    k = np.arange(nz, dtype=np.int64)[:, None, None]
    z = np.asarray(sc["height"], dtype=np.float64)[:, None, None]
    sc["oro"] = 400.0 * (1.0 + _wave(2 * i[0] + 3 * j[0], per)) + 0.0 * j[0]
    pv = np.empty((2, nz, ny, nx)); qv = np.empty((2, nz, ny, nx))
    for m in range(2):
        sh = 10 * m
        pv[m] = 1.0e-6 * (1.0 + 0.5 * _wave(i + sh + k, per)) * (1.0 + z / 6000.0) * (2.0 * j / float(ny - 1) - 1.0)
        qv[m] = 0.012 / (1.0 + z / 2500.0) ** 2 * (0.6 + 0.4 * _wave(2 * (i + sh) + j, per)) + 0.0 * k
    if xcyclic_ok(nx):
        pv[..., nx - 1] = pv[..., 0]; qv[..., nx - 1] = qv[..., 0]; sc["oro"][..., nx - 1] = sc["oro"][..., 0]
    sc["pv"] = pv; sc["qv"] = qv
    it = int(sc.get("itime0", 0)) if itime is None else int(itime)
    sc["itime"] = it
    itra1 = np.full(int(sc["npart"]), it, np.int32)
    if dead_every:
        itra1[::dead_every] = -999999999
    sc["itra1"] = itra1
    sc["npoint"] = (1 + np.arange(int(sc["npart"])) % 3).astype(np.int32)
    return sc


def nest_model_levels(m, ix0=10, jy0=6, ix1=25, jy1=16, factor=2, phase=3):
    """Model-level input on one nested grid covering mother cells [ix0,ix1] x [jy0,jy1] at `factor` times the
    resolution (verttransform_nests.f90); same hybrid coefficients as the mother, its own smooth patterns."""
    nx, ny, nz = (int(v) for v in m["grid"])
    dx, dy, xlon0, ylat0 = (float(v) for v in m["geom"])
    nxn, nyn = (ix1 - ix0) * factor + 1, (jy1 - jy0) * factor + 1
    n = model_levels(nx=nxn, ny=nyn, nz=nz, global_grid=False, polar=False, phase=phase)
    for k in ("akz", "bkz", "aknew", "bknew"):
        n[k] = m[k].copy()
    n["geom"] = np.array([dx / factor, dy / factor, xlon0 + ix0 * dx, ylat0 + jy0 * dy], np.float64)
    n["globalflags"] = np.array([0, 0, 0], np.int32)
    return n


# --------------------------------------------------------------------------
# concoutput: output grids with the run structure the sparse writer compresses
# --------------------------------------------------------------------------
def concoutput_case(nxg=24, nyg=16, nzg=4, nspec=2, wet=True, dry=True, itime=3600, seed=5, classes=1):
    """gridunc / wetgridunc / drygridunc with empty stretches, single cells and long runs (float32 values, as the
    sampling kernels leave them), plus area/volume as outgrid_init.f90:59-95 computes them.  classes > 1: the grids carry
    a leading uncertainty-class dimension [classes][nspec]... (nclassunc of par_mod; concoutput writes the class mean times
    nclassunc, concoutput.f90:296-345 with mean_mod)."""
    if classes > 1:
        parts = [concoutput_case(nxg, nyg, nzg, nspec, wet, dry, itime, seed + 17 * c) for c in range(classes)]
        co = dict(parts[0])
        co["classes"] = np.array([classes], np.int32)
        for k in ("gridunc", "wetgridunc", "drygridunc"):
            if k in co:
                co[k] = np.stack([p[k] for p in parts])
        return co
    u = _uniform01(nspec * nzg * nyg * nxg, seed).reshape(nspec, nzg, nyg, nxg)
    blob = _wave(np.arange(nxg)[None, None, None, :] * 3 + np.arange(nyg)[None, None, :, None] * 5, 4 * nxg)
    g = np.where((u > 0.55) & (blob > -0.2), (u * 1.0e-3).astype(np.float32), np.float32(0.0)).astype(np.float64)
    g[:, :, 0, :] = 0.0                      # a whole empty row
    g[:, 1, 3, :] = 2.0e-4                   # a whole full row
    w2 = np.where(u[:, 0] > 0.7, (u[:, 0] * 1.0e-5).astype(np.float32), np.float32(0.0)).astype(np.float64)
    d2 = np.where(u[:, 1] > 0.4, (u[:, 1] * 1.0e-6).astype(np.float32), np.float32(0.0)).astype(np.float64)
    outheight = np.array([100.0, 500.0, 1000.0, 5000.0, 10000.0, 15000.0, 20000.0, 25000.0, 30000.0, 40000.0, 50000.0, 60000.0][:nzg])
    dxout, dyout, outlon0, outlat0 = 1.0, 1.0, -10.0, 30.0
    f32 = np.float32
    pi, r_earth = f32(3.14159265), f32(6.371e6)
    pi180 = pi / f32(180.0)
    area = np.zeros((nyg, nxg)); volume = np.zeros((nzg, nyg, nxg))
    for jy in range(nyg):                    # outgrid_init.f90:59-95 in the reference's own real kind (f32)
        ylat = f32(outlat0) + (f32(jy) + f32(0.5)) * f32(dyout)
        ylatp = ylat + f32(0.5) * f32(dyout); ylatm = ylat - f32(0.5) * f32(dyout)
        if ylatm < 0 and ylatp > 0:
            hzone = f32(dyout) * r_earth * pi180
        else:
            cp = f32(np.cos(ylatp * pi180)); cm = f32(np.cos(ylatm * pi180))
            a_, b_ = (cp, cm) if cp < cm else (cm, cp)
            hzone = (f32(np.sqrt(f32(1) - a_ * a_)) - f32(np.sqrt(f32(1) - b_ * b_))) * r_earth
        ga = f32(2.0) * pi * r_earth * hzone * f32(dxout) / f32(360.0)
        area[jy, :] = ga
        volume[0, jy, :] = ga * f32(outheight[0])
        for kz in range(1, nzg):
            volume[kz, jy, :] = ga * (f32(outheight[kz]) - f32(outheight[kz - 1]))
    co = dict(outgrid=np.array([nxg, nyg, nzg, nspec, int(wet), int(dry), itime], np.int32),
              outgeom=np.array([dxout, dyout, outlon0, outlat0, 4.0]), outheight=outheight, area=area, volume=volume, gridunc=g)
    if wet:
        co["wetgridunc"] = w2
    if dry:
        co["drygridunc"] = d2
    return co


def add_pptv(co, nx=40, ny=24, nz=12):
    """Mixing-ratio output (iout = 3) for a concoutput_case(): a met grid that covers the output grid, z levels, the air
    density of slot memind(2) and molar weights (concoutput.f90:176-205,482-590)."""
    nxg, nyg = int(co["outgrid"][0]), int(co["outgrid"][1])
    dxout, dyout, outlon0, outlat0 = (float(v) for v in co["outgeom"][:4])
    co["iout"] = 3
    co["met"] = np.array([nx, ny, nz], np.int32)
    co["metgeom"] = np.array([(nxg * dxout + 2.0) / (nx - 1), (nyg * dyout + 2.0) / (ny - 1), outlon0 - 1.0, outlat0 - 1.0])
    co["height"] = make_height(nz, top=30000.0, lin=0.15)
    i = np.arange(nx, dtype=np.int64)[None, None, :]; j = np.arange(ny, dtype=np.int64)[None, :, None]
    z = np.asarray(co["height"])[:, None, None]
    co["rho2"] = (1.2 / (1.0 + z / 8000.0 + 0.5 * (z / 8000.0) ** 2) * (1.0 + 0.02 * _wave(i + 2 * j, nx - 1))).astype(np.float32).astype(np.float64)
    co["weightmolar"] = np.array([350.0, 28.0, 64.0, 100.0, 46.0][: int(co["outgrid"][3])])
    return co


# --------------------------------------------------------------------------
# releaseparticles + particle splitting
# --------------------------------------------------------------------------
def release_case(nx=48, ny=32, nz=24, *, nspec=2, ind_rel=1, mquasilag=0, maxpart=100000, itsplit=1800, existing=400,
                 times=(0, 900, 1800, 2700, 3600), global_grid=True, ibdate=20200615, ibtime=120000, nest=False):
    """RELEASES with five release points of every kind the routine distinguishes: a point source released at one
    instant (kindz 1), a box released over an interval (area source, metres above ground), a box in metres above sea
    level (kindz 2: topography subtracted), a box given in pressure (kindz 3: converted with rho and tt of time slot
    2), and a box across the date line of a global grid (the xglobal wrap).  Particles of a run in progress occupy
    the first storage spaces, some of them terminated (vacant), some due for splitting.  Local-time emission factors
    (point_hour, area_dow ...) differ per species.  nest=True: a nested wind field of twice the resolution over mother cells
    [16,32] x [6,22] with its own orography, density and temperature: the second box straddles its western edge, the box above
    sea level lies inside it, the pressure box straddles its eastern edge (releaseparticles.f90:196-226)."""
    height = make_height(nz, top=30000.0, lin=0.15)
    dx = 360.0 / (nx - 1) if global_grid else 1.0
    dy = 180.0 / (ny - 1) if global_grid else 1.0
    xlon0, ylat0 = (-180.0, -90.0) if global_grid else (-20.0, 20.0)
    f = make_fields(nx, ny, nz, height, nspec=nspec)
    per = nx - 1
    i = np.arange(nx, dtype=np.int64)[None, :]
    j = np.arange(ny, dtype=np.int64)[:, None]
    oro = 300.0 * (1.0 + _wave(2 * i + 3 * j, per)) + 0.0 * j
    oro[:, nx - 1] = oro[:, 0]
    rs = dict(grid=np.array([nx, ny, nz], np.int32), geom=np.array([dx, dy, xlon0, ylat0]), xglobal=int(global_grid),
              height=height, nspec=nspec, bdate=np.array([ibdate, ibtime], np.int32),
              switches=np.array([1, 900, 1, itsplit, ind_rel, mquasilag, maxpart, 1], np.int32),
              times=np.asarray(times, np.int32), oro=oro, rho2=f["rho"][1], tt2=f["tt"][1])
    # release points in grid coordinates (readreleases.f90 converts lon/lat with (x-xlon0)/dx)
    xm = nx - 1.0 if global_grid else nx - 4.0      # the last box crosses the date line of a global grid only
    rs.update(numpoint=5,
              ireleasestart=np.array([0, 0, 900, 0, 900], np.int32), ireleaseend=np.array([0, 3600, 2700, 3600, 900], np.int32),
              npart_rel=np.array([300, 1000, 800, 600, 500], np.int32), kindz=np.array([1, 1, 2, 3, 1], np.int32),
              xpoint1=np.array([10.25, 14.0, 20.5, 30.0, xm - 1.5]), xpoint2=np.array([10.25, 17.5, 24.0, 33.0, xm + 1.5]),
              ypoint1=np.array([12.5, 8.0, 18.0, 10.0, 15.0]), ypoint2=np.array([12.5, 11.0, 21.5, 13.0, 17.0]),
              zpoint1=np.array([50.0, 0.0, 900.0, 850.0, 100.0]), zpoint2=np.array([50.0, 800.0, 2500.0, 500.0, 3000.0]),
              xmass=np.array([[1.0, 5.0, 2.0, 1.5, 0.8], [0.5, 0.0, 4.0, 1.0, 0.3]][:nspec]))
    hour = 1.0 + 0.5 * _wave(np.arange(24)[:, None] * 2 + np.arange(nspec)[None, :] * 5, 24)
    dow = 1.0 + 0.25 * _wave(np.arange(7)[:, None] * 3 + np.arange(nspec)[None, :], 7)
    rs.update(point_hour=hour, area_hour=hour[::-1].copy(), point_dow=dow, area_dow=dow[::-1].copy())
    if nest:
        ix0, ix1, jy0, jy1, fac = 16, 32, 6, 22, 2
        nxn, nyn = fac * (ix1 - ix0) + 1, fac * (jy1 - jy0) + 1
        fn = make_fields(nxn, nyn, nz, height, nspec=nspec)
        gi = np.arange(nxn, dtype=np.int64)[None, :]
        gj = np.arange(nyn, dtype=np.int64)[:, None]
        rs.update(nest=np.array([nxn, nyn], np.int32), nestcorners=np.array([ix0, jy0, ix1, jy1, fac, fac], np.float64), par_nxmax=721,
                  oron=420.0 * (1.0 + _wave(3 * gi + gj, 2 * per)) + 0.0 * gj, rhon2=fn["rho"][1] * 1.03, ttn2=fn["tt"][1] - 2.5)
    if existing:
        n = int(existing)
        u = _uniform01(n, 77)
        itra1 = np.zeros(n, np.int32)                      # all synchronised at the first release time ...
        itra1[::5] = -999999999                            # ... except the terminated ones: vacant storage spaces
        itramem = -(900 * (1 + (np.arange(n) % 4))).astype(np.int32)
        rs.update(npart=n, xtra1=2.0 + u * (nx - 5.0), ytra1=2.0 + _uniform01(n, 78) * (ny - 5.0), ztra1=10.0 + 3000.0 * _uniform01(n, 79),
                  itra1=itra1, itramem=itramem, itrasplit=(itramem + itsplit).astype(np.int32),
                  npoint=np.ones(n, np.int32), nclass=np.ones(n, np.int32), idt=np.full(n, 1, np.int32),
                  uap=_uniform01(n, 80) - 0.5, xmass1=np.ones((nspec, n)) * 0.01)
    return rs


# --------------------------------------------------------------------------
# calcpar: the surface analysis it reads besides the model levels
# --------------------------------------------------------------------------
def calcpar_inputs(m, lsubgrid=1):
    """surfstr, sshf (both signs: stable and convective columns), excessoro and the half-level coefficients akm, bkm for a
    synthetic.model_levels() dict (compact [ny][nx] arrays)."""
    nx, ny, nz = (int(v) for v in m["grid"])
    per = nx - 1
    i = np.arange(nx, dtype=np.int64)[None, :]
    j = np.arange(ny, dtype=np.int64)[:, None]
    surfstr = 0.02 + 0.5 * (0.5 + 0.5 * _wave(3 * i + j, per)) ** 2 + 0.0 * j
    sshf = -180.0 * _wave(2 * i + 3 * j, per) + 15.0 * _wave(5 * i, per) + 0.0 * j      # W/m2, negative = upward (convective)
    sshf = np.where(np.abs(sshf) < 4.0, 0.0, sshf)                                        # some columns with zero heat flux
    exc = 250.0 * np.maximum(0.0, _wave(i + 2 * j, per)) + 0.0 * j
    for a in (surfstr, sshf, exc):
        a[:, nx - 1] = a[:, 0]
    ak, bk = np.asarray(m["akz"]), np.asarray(m["bkz"])
    akm = np.concatenate([[ak[0]], 0.5 * (ak[1:] + ak[:-1])])          # half levels between the full levels (akm(1) = surface)
    bkm = np.concatenate([[bk[0]], 0.5 * (bk[1:] + bk[:-1])])
    return dict(surfstr=surfstr, sshf=sshf, excessoro=exc, akm=akm, bkm=bkm, lsubgrid=int(lsubgrid))


# --------------------------------------------------------------------------
# convective mixing (convmix.f90): soundings with and without CAPE
# --------------------------------------------------------------------------
def convection_case(nx=24, ny=16, nuvz=46, n=4000, ncalls=3, ldirect=1, lsynctime=900, seed=11, nest=False):
    """ECMWF-shaped input of the convection scheme on a small grid: hybrid half levels akm, bkm (akm(1) = surface) with
    the full levels akz, bkz between them (level 1 = surface), two time slots of ps, tt2, td2, tth, qvh [2][nuvz][ny][nx]
    whose soundings range from warm, moist and conditionally unstable (deep convection: most of the matrix is used)
    over shallow and marginal cases to dry or cold columns in which CONVECT leaves through each of its early exits;
    `n` particles between the ground and 16 km, a tenth of them not due, `ncalls` consecutive calls one lsynctime apart
    (the cloud-base mass flux cbaseflux relaxes from call to call).  nest=True adds one nested wind field of twice the
    resolution over the middle of the domain with soundings of its own (a wrong grid choice shows) and its own mass flux."""
    per = nx
    i = np.arange(nx, dtype=np.int64)[None, None, :]
    j = np.arange(ny, dtype=np.int64)[None, :, None]
    k = np.arange(nuvz, dtype=np.float64)
    eta_h = np.exp(-3.3 * (k / float(nuvz - 1)) ** 1.25)                 # half levels: 1 ... about 0.037 (37 hPa)
    bkm = eta_h ** 1.6
    bkm[0] = 1.0
    akm = 101325.0 * (eta_h - bkm)
    akm[0] = 0.0
    akz = np.concatenate([[0.0], 0.5 * (akm[:-1] + akm[1:])])
    bkz = np.concatenate([[1.0], 0.5 * (bkm[:-1] + bkm[1:])])
    out = dict(akz=akz, bkz=bkz, akm=akm, bkm=bkm)
    def soundings(gnx, gny, shift, scale):
        """ps, tt2, td2 [2][gny][gnx], tth, qvh [2][nuvz][gny][gnx]; `scale` grid cells of this grid per mother cell."""
        gi = np.arange(gnx, dtype=np.int64)[None, None, :]
        gj = np.arange(gny, dtype=np.int64)[None, :, None]
        gper = per * scale
        res = {}
        for slot in range(2):
            ii = gi + 3 * slot * scale + shift
            warm = 0.5 + 0.5 * _wave(ii[0] + 2 * gj[0], gper)            # 0 ... 1: cold/dry ... warm/moist
            ps = 101300.0 + 600.0 * _wave(2 * ii[0] + gj[0], gper) - 14000.0 * np.maximum(0.0, _wave(3 * ii[0] + 5 * gj[0], 2 * gper)) ** 3
            p = akz[:, None, None] + bkz[:, None, None] * ps[None]
            z = -7600.0 * np.log(p / ps[None])
            tsfc = 271.0 + 31.0 * warm + 0.0 * ps
            lapse = 0.0055 + 0.0015 * warm
            tth = np.maximum(tsfc[None] - lapse[None] * z, 198.0 + 8.0 * warm[None]) + 0.4 * _wave(ii + gj + 3 * np.arange(nuvz)[:, None, None], gper)
            rh = (0.25 + 0.65 * warm[None]) * np.exp(-z / (5000.0 + 4000.0 * warm[None])) + 0.05
            tc = tth - 273.15
            es = 611.2 * np.exp(17.67 * tc / (tc + 243.5))
            qs = 0.622 * es / np.maximum(p - 0.378 * es, 1.0)
            qvh = np.minimum(rh, 0.98) * qs
            tt2 = tth[0] + 0.6
            td2 = tt2 - (2.0 + 14.0 * (1.0 - warm))
            res[slot] = (ps, tt2, td2, tth, qvh)
        return {key: np.stack([res[0][idx], res[1][idx]]) for idx, key in enumerate(("ps", "tt2", "td2", "tth", "qvh"))}

    out.update(soundings(nx, ny, 0, 1))
    if nest:
        ix0, jy0, ix1, jy1, fac = nx // 4, ny // 4, (2 * nx) // 3, (2 * ny) // 3, 2
        nxn, nyn = (ix1 - ix0) * fac + 1, (jy1 - jy0) * fac + 1
        sn = soundings(nxn, nyn, 5, fac)
        out.update({k + "n": v for k, v in sn.items()})
        gi = np.arange(nxn, dtype=np.int64)[None, :]
        gj = np.arange(nyn, dtype=np.int64)[:, None]
        warmn = 0.5 + 0.5 * _wave(gi + 5 + 2 * gj, per * fac)
        # xln, yln, xrn, yrn in mother grid units, xresoln, yresoln (gridcheck_nests.f90)
        out.update(nest=np.array([nxn, nyn], np.int32), nestgeom=np.array([ix0, jy0, ix1, jy1, fac, fac], np.float64), par_nxmax=721,
                   cbasefluxn=np.where(warmn > 0.5, 0.1 * warmn ** 2, 0.0))
    # a run in progress: the cloud-base mass flux of the warm columns has built up (it starts from zero in others)
    warm0 = 0.5 + 0.5 * _wave(i[0] + 2 * j[0], per)
    cb0 = np.where(warm0 > 0.55, 0.12 * warm0 ** 2, 0.0)
    u = [_splitmix64(n, seed + q).astype(np.float64) / 2.0 ** 64 for q in range(4)]
    # backward runs: lsynctime and the wind-field times carry the sign of ldirect (readcommand.f90:381-384, getfields.f90)
    out.update(grid=np.array([nx, ny, nuvz], np.int32), nconvlev=nuvz - 2, ldirect=int(ldirect), lsynctime=int(lsynctime) * int(ldirect),
               memtime=np.array([0, 10800 * int(ldirect)], np.int32), itimes=np.arange(ncalls, dtype=np.int32) * int(lsynctime) * int(ldirect),
               height_nz=19000.0, cbaseflux=cb0,
               xtra1=0.3 + u[0] * (nx - 1.6), ytra1=0.3 + u[1] * (ny - 1.6), ztra1=16000.0 * u[2] ** 1.5,
               due=(u[3][:, None] + 0.013 * np.arange(ncalls)[None, :]) % 1.0 > 0.1, npart=n)
    return out
