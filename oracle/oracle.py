"""TEST INFRASTRUCTURE ONLY -- ctypes front end of the CPU oracle (oracle/flexpart_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under flexpart_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    """Compile liboracle_r4.so / liboracle_r8.so with gcc (a few seconds)."""
    for stem, lib in (("flexpart_oracle", "liboracle"), ("verttransform_oracle", "libvtoracle"),
                      ("partoutput_oracle", "libpooracle"), ("readpart_oracle", "librporacle"), ("release_oracle", "librloracle"),
                      ("calcpar_oracle", "libcporacle"), ("convect_oracle", "libcvoracle")):
        src = os.path.join(HERE, stem + ".c")
        for kind, real in (("r4", "float"), ("r8", "double")):
            out = os.path.join(HERE, f"{lib}_{kind}.so")
            if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
                subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
                                       f"-DORC_REAL={real}", src, "-o", out, "-lm"])
    _build_co()


def _build_co():
    src = os.path.join(HERE, "concoutput_oracle.c")
    out = os.path.join(HERE, "libcooracle_r4.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", out, "-lm"])
    return out


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class Oracle:
    """The reference's serial particle loop, restated in C, driven from a scenario dict."""

    def __init__(self, sc, kind="r8"):
        build()
        self.kind = kind
        self.rt = np.float32 if kind == "r4" else np.float64
        lib = C.CDLL(os.path.join(HERE, f"liboracle_{kind}.so"))
        self.lib = lib
        lib.orc_create.restype = C.c_void_p
        lib.orc_rannumb.restype = C.c_void_p
        lib.orc_step.restype = C.c_long
        lib.orc_nan_count.restype = C.c_long
        lib.orc_ran3_next.restype = C.c_double
        self.h = C.c_void_p(lib.orc_create())
        assert lib.orc_real_size() == np.dtype(self.rt).itemsize
        self.sc = sc
        nx, ny, nz = (int(v) for v in sc["grid"])
        dx, dy, xlon0, ylat0 = (float(v) for v in sc["geom"])
        xg, ng, sg = (int(v) for v in sc["globalflags"])
        hgt = _f64(sc["height"])
        dp = C.POINTER(C.c_double)
        lib.orc_set_grid(self.h, nx, ny, nz, C.c_double(dx), C.c_double(dy), C.c_double(xlon0),
                         C.c_double(ylat0), xg, ng, sg, hgt.ctypes.data_as(dp), int(sc["nmixz"]))
        if "northpolemap" in sc:
            n = _f64(sc["northpolemap"]); s = _f64(sc["southpolemap"])
            lib.orc_set_polemaps(self.h, n.ctypes.data_as(dp), s.ctypes.data_as(dp))
        elif ng or sg:
            lib.orc_make_polemaps(self.h)
        mt = sc["memtime"]; mi = sc["memind"]
        lib.orc_set_time(self.h, int(mt[0]), int(mt[1]), int(mi[0]), int(mi[1]))
        nspec = int(sc["nspec"])
        dds = np.ascontiguousarray(np.asarray(sc["drydepspec"], dtype=np.int32))
        tp = sc["turbpar"]
        lib.orc_set_switches(self.h, int(sc["ldirect"]), int(sc["lsynctime"]), int(sc["method"]),
                             int(sc["mintime"]), C.c_double(float(sc["ctl"])), int(sc["ifine"]),
                             int(sc["turbswitch"]), int(sc["cblflag"]), int(sc["mdomainfill"]),
                             int(sc["lsettling"]), nspec, int(sc["drydep"]),
                             dds.ctypes.data_as(C.POINTER(C.c_int)), C.c_double(float(tp[0])),
                             C.c_double(float(tp[1])), C.c_double(float(tp[2])))
        lib.orc_set_com_parameters(self.h, int(sc.get("turboff", 0)), int(sc.get("interpolhmix", 0)))   # com_mod.f90:777-778
        arrs = [_f64(sc[k]) for k in ("density", "dquer", "vsetaver", "cunningham", "decay")]
        lib.orc_set_species(self.h, *[a.ctypes.data_as(dp) for a in arrs],
                            int(np.asarray(sc["lage"]).ravel()[-1]), int(sc.get("mquasilag", 0)))
        # release-point tables: xmass [nspec][numpoint], npart_rel [numpoint]; default = one point of unit mass
        numpoint = int(sc.get("numpoint", 1))
        self.numpoint = numpoint
        xm = _f64(np.asarray(sc.get("xmass", np.ones(nspec * numpoint)), dtype=np.float64).reshape(nspec, numpoint))
        npt = np.ascontiguousarray(np.asarray(sc.get("npart_rel", np.full(numpoint, max(int(sc["npart"]), 1))), dtype=np.int32).ravel())
        lib.orc_set_release_points(self.h, numpoint, xm.ctypes.data_as(dp), npt.ctypes.data_as(C.POINTER(C.c_int)))
        # fields in the oracle's precision (kept alive on self)
        self.f = {}
        ptrs = []
        for k in ("uu", "vv", "ww", "rho", "drhodz", "tt", "uupol", "vvpol", "hmix", "ustar",
                  "wstar", "oli", "tropopause", "vdep"):
            if k in sc:
                self.f[k] = np.ascontiguousarray(np.asarray(sc[k]).astype(self.rt))
                ptrs.append(self.f[k].ctypes.data_as(C.c_void_p))
            else:
                ptrs.append(C.c_void_p(0))
        lib.orc_set_fields(self.h, *ptrs)
        lib.orc_set_eps_nxmax(self.h, int(sc.get("par_nxmax", 361)))
        if "nest" in sc:
            nxn, nyn = (int(v) for v in sc["nest"])
            dxn, dyn, lon0n, lat0n = (float(v) for v in sc["nestgeom"])
            self.nf = {}
            nptrs = []
            for k in ("uun", "vvn", "wwn", "rhon", "drhodzn", "hmixn", "ustarn", "wstarn", "olin", "tropopausen", "vdepn"):
                self.nf[k] = np.ascontiguousarray(np.asarray(sc[k]).astype(self.rt))
                nptrs.append(self.nf[k].ctypes.data_as(C.c_void_p))
            lib.orc_set_nest(self.h, nxn, nyn, C.c_double(dxn), C.c_double(dyn), C.c_double(lon0n), C.c_double(lat0n), *nptrs)
        # particle state
        n = int(sc["npart"])
        self.n = n
        self.nspec = nspec
        self.x = _f64(sc["xtra1"]).copy()
        self.y = _f64(sc["ytra1"]).copy()
        self.z = np.asarray(sc["ztra1"]).astype(self.rt)

        def opt(name, dtype, fill=0):
            if name in sc:
                return np.ascontiguousarray(np.asarray(sc[name]).astype(dtype))
            return np.full(n, fill, dtype)
        self.uap = opt("uap", self.rt); self.ucp = opt("ucp", self.rt); self.uzp = opt("uzp", self.rt)
        self.us = opt("us", self.rt); self.vs = opt("vs", self.rt); self.ws = opt("ws", self.rt)
        self.idt = opt("idt", np.int32); self.itra1 = opt("itra1", np.int32)
        self.itramem = opt("itramem", np.int32); self.npoint = opt("npoint", np.int32, 1)
        self.cbt = opt("cbt", np.int16, 1)
        self.xmass1 = np.ascontiguousarray(np.asarray(sc["xmass1"]).astype(self.rt).reshape(nspec, n))
        self.prob = np.zeros((nspec, n), self.rt)
        self.itime = int(sc["itime0"])
        self.nclass = opt("nclass", np.int32, 1)
        self.has_grid = "outgrid" in sc
        if self.has_grid:
            nxg, nyg, nzg = (int(v) for v in sc["outgrid"])
            dxo, dyo, lon0, lat0 = (float(v) for v in sc["outgeom"])
            oh = _f64(sc["outheight"])
            lage = np.ascontiguousarray(np.asarray(sc["lage"], dtype=np.int32).ravel())
            ind_samp, iofr = (int(v) for v in sc["concflags"])
            mps = self.numpoint if iofr == 1 else 1          # maxpointspec_act
            ncu = int(sc.get("nclassunc", 1))
            self.gshape = (len(lage), ncu, mps, nspec, nzg, nyg, nxg)   # (nage, nclassunc, maxpointspec, spec, z, y, x)
            lib.orc_set_outgrid(self.h, nxg, nyg, nzg, C.c_double(dxo), C.c_double(dyo), C.c_double(lon0),
                                C.c_double(lat0), oh.ctypes.data_as(dp), mps, ncu, len(lage),
                                lage.ctypes.data_as(C.POINTER(C.c_int)), ind_samp, iofr, int(sc.get("lusekerneloutput", 1)), nspec)
            if "outtimes" in sc:
                lib.orc_set_output_times(self.h, int(sc["outtimes"][0]), int(sc["outtimes"][1]))
            self.has_grid_nest = "outgridn" in sc
            if self.has_grid_nest:
                nxn, nyn = (int(v) for v in sc["outgridn"])
                dxn, dyn, lon0n, lat0n = (float(v) for v in sc["outgeomn"])
                self.gshape_nest = (len(lage), ncu, mps, nspec, nzg, nyn, nxn)
                lib.orc_set_outgrid_nest(self.h, nxn, nyn, C.c_double(dxn), C.c_double(dyn), C.c_double(lon0n), C.c_double(lat0n))
            self.nreceptor = 0
            if "receptors" in sc:
                r = _f64(sc["receptors"]).reshape(3, -1)
                self.nreceptor = r.shape[1]
                rx, ry, ra = (np.ascontiguousarray(r[k]) for k in range(3))
                lib.orc_set_receptors(self.h, self.nreceptor, rx.ctypes.data_as(dp), ry.ctypes.data_as(dp), ra.ctypes.data_as(dp))
        self.has_wet = bool(sc.get("wetdep", 0))
        if self.has_wet:
            self._init_wet(sc)
        # backward runs with receptor scavenging (DRYBKDEP / WETBKDEP, timemanager.f90:564-598): xscav_frac1 starts at -1
        # (releaseparticles.f90:167-171) unless the scenario carries the array
        self.bkdep = (int(sc.get("drybkdep", 0)), int(sc.get("wetbkdep", 0)))
        self.xscav = None
        if any(self.bkdep):
            z1 = _f64(sc.get("zpoint1", np.zeros(self.numpoint)))
            z2 = _f64(sc.get("zpoint2", np.zeros(self.numpoint)))
            lib.orc_set_bkdep(self.h, self.bkdep[0], self.bkdep[1], self.numpoint, z1.ctypes.data_as(dp), z2.ctypes.data_as(dp))
            if "xscav_frac1" in sc:
                self.xscav = np.ascontiguousarray(np.asarray(sc["xscav_frac1"]).astype(self.rt).reshape(nspec, n))
            else:
                self.xscav = np.full((nspec, n), -1.0, self.rt)

    def _init_wet(self, sc):
        lib = self.lib
        dp = C.POINTER(C.c_double)
        nspec = self.nspec
        self.wf = {k: np.ascontiguousarray(np.asarray(sc[k]).astype(self.rt)) for k in ("lsprec", "convprec", "tcc")}
        self.wf["clouds"] = np.ascontiguousarray(np.asarray(sc["clouds"]).astype(np.int8))
        self.wf["cloudsh"] = np.ascontiguousarray(np.asarray(sc["cloudsh"]).astype(np.int32))
        par = [_f64(sc[k]) for k in ("weta_gas", "wetb_gas", "crain_aero", "csnow_aero", "ccn_aero", "in_aero", "henry")]
        wds = np.ascontiguousarray(np.asarray(sc["wetdepspec"], dtype=np.int32))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        lib.orc_set_wet(self.h, wds.ctypes.data_as(C.POINTER(C.c_int)), *[a.ctypes.data_as(dp) for a in par], 0,
                        vp(self.wf["lsprec"]), vp(self.wf["convprec"]), vp(self.wf["tcc"]), vp(self.wf["clouds"]),
                        vp(self.wf["cloudsh"]), C.c_void_p(0))
        self._wetpar = (par, wds)
        if "lsprecn" in sc:   # the nest's own precipitation / cloud / temperature fields
            for k in ("lsprecn", "convprecn", "tccn", "ttn"):
                self.wf[k] = np.ascontiguousarray(np.asarray(sc[k]).astype(self.rt))
            self.wf["cloudsn"] = np.ascontiguousarray(np.asarray(sc["cloudsn"]).astype(np.int8))
            lib.orc_set_wet_nest(self.h, vp(self.wf["lsprecn"]), vp(self.wf["convprecn"]), vp(self.wf["tccn"]),
                                 vp(self.wf["cloudsn"]), vp(self.wf["ttn"]))

    def wetgrid(self):
        nage, ncu, mps, nsp, nzg, nyg, nxg = self.gshape
        self.lib.orc_wetgridunc.restype = C.c_void_p
        lead = self._lead(self.gshape)
        d = np.ctypeslib.as_array(C.cast(self.lib.orc_wetgridunc(self.h), C.POINTER(C.c_float)), shape=(int(np.prod(lead)) * nsp * nyg * nxg,))
        return d.astype(np.float64).reshape(lead + (nsp, nyg, nxg))

    @staticmethod
    def _lead(gshape):
        """Leading (age, class, pointspec) extents of the reference's grids; dropped when all are 1 (the single-class
        fixtures of round 1 compare (spec, [z,] y, x) arrays)."""
        return () if gshape[0] * gshape[1] * gshape[2] == 1 else tuple(gshape[:3])

    def grids(self):
        """(gridunc, drygridunc) as float64 arrays shaped (spec, z, y, x) / (spec, y, x)."""
        nage, ncu, mps, nsp, nzg, nyg, nxg = self.gshape
        self.lib.orc_gridunc.restype = C.c_void_p
        self.lib.orc_drygridunc.restype = C.c_void_p
        ct = C.c_float if self.kind == "r4" else C.c_double
        lead = self._lead(self.gshape)
        nl = int(np.prod(lead))
        g = np.ctypeslib.as_array(C.cast(self.lib.orc_gridunc(self.h), C.POINTER(ct)), shape=(nl * nsp * nzg * nyg * nxg,))
        d = np.ctypeslib.as_array(C.cast(self.lib.orc_drygridunc(self.h), C.POINTER(C.c_float)), shape=(nl * nsp * nyg * nxg,))
        return (g.astype(np.float64).reshape(lead + (nsp, nzg, nyg, nxg)), d.astype(np.float64).reshape(lead + (nsp, nyg, nxg)))

    def grids_nest(self):
        """(griduncn, drygriduncn, wetgriduncn) of the nested output grid, float64, (spec, z, y, x) / (spec, y, x)."""
        nage, ncu, mps, nsp, nzg, nyg, nxg = self.gshape_nest
        ct = C.c_float if self.kind == "r4" else C.c_double
        out = []
        lead = self._lead(self.gshape_nest)
        for fn, t, shp in (("orc_griduncn", ct, lead + (nsp, nzg, nyg, nxg)), ("orc_drygriduncn", C.c_float, lead + (nsp, nyg, nxg)),
                           ("orc_wetgriduncn", C.c_float, lead + (nsp, nyg, nxg))):
            f = getattr(self.lib, fn)
            f.restype = C.c_void_p
            a = np.ctypeslib.as_array(C.cast(f(self.h), C.POINTER(t)), shape=(int(np.prod(shp)),))
            out.append(a.astype(np.float64).reshape(shp))
        return tuple(out)

    def receptors(self):
        """creceptor as float64 (spec, receptor)."""
        out = np.zeros(self.nspec * self.nreceptor)
        self.lib.orc_get_receptors(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out.reshape(self.nspec, self.nreceptor)

    def polemaps(self):
        n = np.zeros(9); s = np.zeros(9)
        dp = C.POINTER(C.c_double)
        self.lib.orc_get_polemaps(self.h, n.ctypes.data_as(dp), s.ctypes.data_as(dp))
        return n, s

    def rannumb(self):
        p = self.lib.orc_rannumb(self.h)
        ct = C.c_float if self.kind == "r4" else C.c_double
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(1000000,)).copy()

    def ran3_words(self):
        """The shared ran3 state as 60 int32 words (layout of conv_oracle's ran3_words; [59], redist's own seed, is not ours)."""
        w = np.zeros(60, np.int32)
        self.lib.orc_get_ran3_state(self.h, w.ctypes.data_as(C.c_void_p))
        return w

    def set_ran3_words(self, w):
        w = np.ascontiguousarray(w, dtype=np.int32)
        self.lib.orc_set_ran3_state(self.h, w.ctypes.data_as(C.c_void_p))

    def step(self):
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        if self.has_wet and self.itime != 0:   # timemanager.f90:164-169
            loutnext = int(self.sc["outtimes"][0]) if "outtimes" in self.sc else 0
            self.lib.orc_wetdepo(self.h, self.itime, int(self.sc["lsynctime"]), loutnext, self.n, vp(self.x), vp(self.y),
                                 vp(self.z), vp(self.itra1), vp(self.itramem), vp(self.npoint), vp(self.nclass), vp(self.xmass1))
        nadv = self.lib.orc_step(self.h, self.itime, self.n, vp(self.x), vp(self.y), vp(self.z),
                                 vp(self.uap), vp(self.ucp), vp(self.uzp), vp(self.us), vp(self.vs),
                                 vp(self.ws), vp(self.idt), vp(self.itra1), vp(self.itramem),
                                 vp(self.npoint), vp(self.cbt), vp(self.xmass1), vp(self.prob), vp(self.nclass), self._xscav_ptr())
        self.itime += int(self.sc["lsynctime"])
        if self.has_grid:   # sample at the new positions (conccalc.f90), weight 1
            self.lib.orc_conccalc(self.h, self.itime, C.c_double(1.0), self.n, vp(self.x), vp(self.y), vp(self.z),
                                  vp(self.itra1), vp(self.itramem), vp(self.npoint), vp(self.nclass), vp(self.xmass1), self._xscav_ptr())
        return nadv

    def _xscav_ptr(self):
        return self.xscav.ctypes.data_as(C.c_void_p) if self.xscav is not None else C.c_void_p(0)

    def sample(self, weight=1.0):
        """conccalc at the current time and positions."""
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        self.lib.orc_conccalc(self.h, self.itime, C.c_double(weight), self.n, vp(self.x), vp(self.y), vp(self.z),
                              vp(self.itra1), vp(self.itramem), vp(self.npoint), vp(self.nclass), vp(self.xmass1), self._xscav_ptr())

    def state(self):
        return dict(xtra1=self.x.copy(), ytra1=self.y.copy(), ztra1=self.z.astype(np.float64),
                    uap=self.uap.astype(np.float64), ucp=self.ucp.astype(np.float64),
                    uzp=self.uzp.astype(np.float64), us=self.us.astype(np.float64),
                    vs=self.vs.astype(np.float64), ws=self.ws.astype(np.float64),
                    idt=self.idt.copy(), itra1=self.itra1.copy(), cbt=self.cbt.astype(np.int32),
                    xmass1=self.xmass1.astype(np.float64), prob=self.prob.astype(np.float64),
                    **({"xscav_frac1": self.xscav.astype(np.float64)} if self.xscav is not None else {}))

    def run(self, nsteps=None):
        out = []
        for _ in range(int(self.sc["nsteps"]) if nsteps is None else nsteps):
            self.step()
            out.append(self.state())
        return out

    def clear_gridunc(self):
        """gridunc, griduncn, creceptor = 0 as concoutput.f90:719-720 leaves them; the deposition grids accumulate on."""
        self.lib.orc_clear_gridunc(self.h)

    def track_leaks(self):
        """Start recording which particles the serial code's two order-dependent leaks touch:
        returns a uint8 array the oracle ORs 1 (advance.f90:550 taken) / 2 (initialize() with the predecessor's
        polar-or-not wind choice) into, per particle number, over all following steps."""
        self._leaks = np.zeros(self.n, np.uint8)
        self.lib.orc_set_leak_flags(self.h, self._leaks.ctypes.data_as(C.c_void_p))
        return self._leaks

    def nan_counts(self):
        return self.lib.orc_nan_count(self.h, 1), self.lib.orc_nan_count(self.h, 2)

    def __del__(self):
        try:
            self.lib.orc_destroy(self.h)
        except Exception:
            pass


def compare(a, b, keys=("xtra1", "ytra1", "ztra1", "uap", "ucp", "uzp", "us", "vs", "ws")):
    """-> dict key -> (max abs diff, max rel-to-range diff), plus integer mismatches."""
    rep = {}
    for k in keys:
        d = np.abs(np.asarray(a[k], np.float64) - np.asarray(b[k], np.float64))
        scale = max(float(np.max(np.abs(b[k]))), 1e-30)
        rep[k] = (float(d.max()), float(d.max() / scale))
    for k in ("idt", "itra1", "cbt"):
        rep[k] = int(np.count_nonzero(np.asarray(a[k]) != np.asarray(b[k])))
    return rep


# --------------------------------------------------------------------------
# verttransform_ecmwf (oracle/verttransform_oracle.c)
# --------------------------------------------------------------------------
VT_OUT = ("uu", "vv", "ww", "tt", "qv", "pv", "rho", "drhodz", "uupol", "vvpol")


class _VtoArgs(C.Structure):
    _dp = C.POINTER(C.c_double)
    _fields_ = ([("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                 ("dx", C.c_double), ("dy", C.c_double), ("xlon0", C.c_double), ("ylat0", C.c_double),
                 ("nglobal", C.c_int), ("sglobal", C.c_int),
                 ("northpolemap", C.c_double * 9), ("southpolemap", C.c_double * 9),
                 ("switchnorthg", C.c_double), ("switchsouthg", C.c_double), ("init", C.c_int),
                 ("cos_dy", C.c_double), ("cos_ylat0", C.c_double), ("xres", C.c_double), ("yres", C.c_double)]
                + [(k, C.POINTER(C.c_double)) for k in ("akz", "bkz", "aknew", "bknew", "ps", "tt2", "td2",
                                                         "tth", "qvh", "uuh", "vvh", "pvh", "wwh", "height")]
                + [("nmixz", C.POINTER(C.c_int))]
                + [(k, C.POINTER(C.c_double)) for k in VT_OUT])


def polemaps(dx, dy, ylat0, kind="r8"):
    """northpolemap, southpolemap, switchnorthg, switchsouthg as gridcheck_ecmwf.f90:341-366 sets them
    (through the restated stlmbr/stcm2p of flexpart_oracle.c)."""
    build()
    lib = C.CDLL(os.path.join(HERE, f"liboracle_{kind}.so"))
    lib.orc_create.restype = C.c_void_p
    h = C.c_void_p(lib.orc_create())
    lib.orc_set_dy_for_polemaps.argtypes = [C.c_void_p, C.c_double]
    lib.orc_set_dy_for_polemaps(h, float(dy))
    lib.orc_make_polemaps.argtypes = [C.c_void_p]
    lib.orc_make_polemaps(h)
    n = np.zeros(9); s = np.zeros(9)
    dp = C.POINTER(C.c_double)
    lib.orc_get_polemaps.argtypes = [C.c_void_p, dp, dp]
    lib.orc_get_polemaps(h, n.ctypes.data_as(dp), s.ctypes.data_as(dp))
    rt = np.float32 if kind == "r4" else np.float64
    swn = float((rt(75.0) - rt(ylat0)) / rt(dy))
    sws = float((rt(-75.0) - rt(ylat0)) / rt(dy))
    return n, s, swn, sws


def vt_oracle(m, kind="r8", height=None, nest_of=None):
    """Run the C restatement of verttransform_ecmwf on a synthetic.model_levels() dict.
    height=None: first call (computes height and nmixz); else the given z levels are used."""
    build()
    lib = C.CDLL(os.path.join(HERE, f"libvtoracle_{kind}.so"))
    nx, ny, nz = (int(v) for v in m["grid"])
    dx, dy, xlon0, ylat0 = (float(v) for v in m["geom"])
    a = _VtoArgs()
    a.nx, a.ny, a.nz = nx, ny, nz
    a.dx, a.dy, a.xlon0, a.ylat0 = dx, dy, xlon0, ylat0
    a.xres = a.yres = 1.0
    a.cos_dy, a.cos_ylat0 = dy, ylat0
    if nest_of is not None:
        # verttransform_nests.f90: the nest's own dyn/ylat0n in cosf (:346), the MOTHER's dxconst/dyconst times
        # xresoln = dx/dxn, yresoln = dy/dyn (gridcheck_nests.f90:359-360, in the host's real kind) in the slope term (:384-385)
        rt = np.float32 if kind == "r4" else np.float64
        mdx, mdy = (rt(v) for v in nest_of["geom"][:2])
        a.xres = float(mdx / rt(dx)); a.yres = float(mdy / rt(dy))
        a.dx, a.dy = float(mdx), float(mdy)
    a.nglobal, a.sglobal = int(m["globalflags"][1]), int(m["globalflags"][2])
    a.switchnorthg = a.switchsouthg = 999999.0
    if a.nglobal or a.sglobal:
        n, s, swn, sws = polemaps(dx, dy, ylat0, kind)
        for i in range(9):
            a.northpolemap[i] = n[i]; a.southpolemap[i] = s[i]
        if a.nglobal: a.switchnorthg = swn
        if a.sglobal: a.switchsouthg = sws
    dp = C.POINTER(C.c_double)
    keep = {}
    for k in ("akz", "bkz", "aknew", "bknew", "ps", "tt2", "td2", "tth", "qvh", "uuh", "vvh", "pvh", "wwh"):
        keep[k] = _f64(m[k])
        setattr(a, k, keep[k].ctypes.data_as(dp))
    h = np.zeros(nz) if height is None else _f64(height).copy()
    a.init = 1 if height is None else 0
    a.height = h.ctypes.data_as(dp)
    nmixz = C.c_int(0)
    a.nmixz = C.pointer(nmixz)
    out = {}
    for k in VT_OUT:
        out[k] = np.zeros((nz, ny, nx))
        setattr(a, k, out[k].ctypes.data_as(dp))
    rc = lib.vto_verttransform(C.byref(a))
    if rc != 0:
        raise RuntimeError(f"vto_verttransform failed ({rc})")
    out["height"] = h
    out["nmixz"] = nmixz.value
    return out


# --------------------------------------------------------------------------
# partoutput (oracle/partoutput_oracle.c)
# --------------------------------------------------------------------------
class _PooArgs(C.Structure):
    _fields_ = ([(k, C.c_int) for k in ("nx", "ny", "nz", "nymax", "nspec", "itime")]
                + [("memtime", C.c_int * 2), ("memind", C.c_int * 2)]
                + [(k, C.c_double) for k in ("dx", "dy", "xlon0", "ylat0")]
                + [(k, C.POINTER(C.c_double)) for k in ("height", "oro", "pv", "qv", "tt", "rho", "hmix", "tropopause")]
                + [("numpart", C.c_long)]
                + [(k, C.POINTER(C.c_double)) for k in ("xtra1", "ytra1", "ztra1")]
                + [(k, C.POINTER(C.c_int)) for k in ("itra1", "itramem", "npoint")]
                + [("xmass1", C.POINTER(C.c_double))])


def po_oracle(sc, kind="r8", nymax=None):
    """File image of partposit_* for a scenario (synthetic.add_partoutput_fields), as bytes."""
    build()
    lib = C.CDLL(os.path.join(HERE, f"libpooracle_{kind}.so"))
    lib.poo_partoutput.restype = C.c_long
    nx, ny, nz = (int(v) for v in sc["grid"])
    a = _PooArgs()
    a.nx, a.ny, a.nz = nx, ny, nz
    a.nymax = int(nymax or 181)            # par_mod.f90:144 of the reference build the fixtures come from
    a.nspec = int(sc["nspec"]); a.itime = int(sc["itime"])
    a.memtime[0], a.memtime[1] = int(sc["memtime"][0]), int(sc["memtime"][1])
    a.memind[0], a.memind[1] = int(sc["memind"][0]), int(sc["memind"][1])
    a.dx, a.dy, a.xlon0, a.ylat0 = (float(v) for v in sc["geom"])
    keep = {}
    dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int)
    for k in ("height", "oro", "pv", "qv", "tt", "rho", "hmix", "tropopause", "xtra1", "ytra1", "ztra1"):
        keep[k] = _f64(sc[k]); setattr(a, k, keep[k].ctypes.data_as(dp))
    for k in ("itra1", "itramem", "npoint"):
        keep[k] = np.ascontiguousarray(np.asarray(sc[k], dtype=np.int32)); setattr(a, k, keep[k].ctypes.data_as(ip))
    n = int(sc["npart"])
    a.numpart = n
    keep["xmass1"] = _f64(np.asarray(sc["xmass1"]).reshape(a.nspec, n)); a.xmass1 = keep["xmass1"].ctypes.data_as(dp)
    cap = 12 + (n + 1) * (16 + (10 + a.nspec) * 8)
    buf = (C.c_ubyte * cap)()
    nb = lib.poo_partoutput(C.byref(a), buf, cap)
    if nb < 0:
        raise RuntimeError("poo_partoutput: buffer too small")
    return bytes(buf[:nb])


# --------------------------------------------------------------------------
# readpartpositions (oracle/readpart_oracle.c)
# --------------------------------------------------------------------------
class _RpoArgs(C.Structure):
    _fields_ = ([(k, C.c_int) for k in ("nspec", "ldirect", "mintime", "itsplit", "nclassunc", "ibdatein", "ibtimein")]
                + [(k, C.c_double) for k in ("bdate", "dx", "dy", "xlon0", "ylat0")]
                + [("maxpart", C.c_long), ("numpart", C.c_long)]
                + [(k, C.c_int) for k in ("numparticlecount", "itimein", "status")]
                + [(k, C.POINTER(C.c_double)) for k in ("xtra1", "ytra1", "ztra1", "xmass1")]
                + [(k, C.POINTER(C.c_int)) for k in ("npoint", "itramem", "nclass", "idt", "itra1", "itrasplit")])


def juldate(yyyymmdd, hhmiss, kind="r8"):
    build()
    lib = C.CDLL(os.path.join(HERE, f"librporacle_{kind}.so"))
    lib.rpo_juldate.restype = C.c_double
    return lib.rpo_juldate(int(yyyymmdd), int(hhmiss))


def rp_oracle(file_bytes, rs, kind="r8"):
    """Warm start from a dump: rs = dict(geom, nspec, restart=[ibdate, ibtime, ibdatein, ibtimein, ldirect, mintime,
    itsplit, nclassunc, numpoint, maxpart]) -> dict of the particle arrays readpartpositions fills."""
    build()
    lib = C.CDLL(os.path.join(HERE, f"librporacle_{kind}.so"))
    r = [int(v) for v in rs["restart"]]
    a = _RpoArgs()
    a.nspec = int(rs["nspec"]); a.ldirect = r[4]; a.mintime = r[5]; a.itsplit = r[6]; a.nclassunc = r[7]
    a.ibdatein, a.ibtimein = r[2], r[3]
    a.bdate = juldate(r[0], r[1], kind)
    a.dx, a.dy, a.xlon0, a.ylat0 = (float(v) for v in rs["geom"])
    mp = r[9]
    a.maxpart = mp
    out = {k: np.zeros(mp) for k in ("xtra1", "ytra1", "ztra1")}
    out["xmass1"] = np.zeros((a.nspec, mp))
    for k in ("npoint", "itramem", "nclass", "idt", "itra1", "itrasplit"):
        out[k] = np.zeros(mp, np.int32)
    dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int)
    for k, v in out.items():
        setattr(a, k, v.ctypes.data_as(dp if v.dtype == np.float64 else ip))
    buf = (C.c_ubyte * len(file_bytes)).from_buffer_copy(file_bytes)
    rc = lib.rpo_readpartpositions(C.byref(a), buf, len(file_bytes))
    if rc != 0:
        raise RuntimeError("rpo_readpartpositions: malformed dump")
    n = int(a.numpart)
    res = {k: (v[:, :n] if v.ndim == 2 else v[:n]).copy() for k, v in out.items()}
    res.update(numpart=n, numparticlecount=int(a.numparticlecount), itimein=int(a.itimein), status=int(a.status))
    return res


# --------------------------------------------------------------------------
# concoutput (oracle/concoutput_oracle.c; float only, like the reference)
# --------------------------------------------------------------------------
class _CooArgs(C.Structure):
    _fields_ = ([(k, C.c_int) for k in ("nxg", "nyg", "nzg", "nspec", "nclassunc", "wetdep", "drydep", "itime")]
                + [("outnum", C.c_float)]
                + [(k, C.POINTER(C.c_float)) for k in ("area", "volume", "gridunc", "wetgridunc", "drygridunc")]
                + [(k, C.c_int) for k in ("nx", "ny", "nz")]
                + [(k, C.c_float) for k in ("dx", "dy", "xlon0", "ylat0", "dxout", "dyout", "outlon0", "outlat0")]
                + [(k, C.POINTER(C.c_float)) for k in ("height", "outheight", "rho", "weightmolar")])


def co_oracle(co):
    """{file name suffix _<species>: bytes} of the grid_conc files for a synthetic.concoutput_case() dict."""
    lib = C.CDLL(_build_co())
    lib.coo_concoutput.restype = C.c_long
    nxg, nyg, nzg, nspec, wet, dry, itime = (int(v) for v in co["outgrid"])
    a = _CooArgs()
    a.nxg, a.nyg, a.nzg, a.nspec, a.nclassunc, a.wetdep, a.drydep, a.itime = nxg, nyg, nzg, nspec, int(np.asarray(co.get("classes", 1)).ravel()[0]), wet, dry, itime
    a.outnum = float(co["outgeom"][4])
    keep = {}
    fp = C.POINTER(C.c_float)
    for k in ("area", "volume", "gridunc", "wetgridunc", "drygridunc"):
        keep[k] = np.ascontiguousarray(np.asarray(co.get(k, np.zeros(1)), dtype=np.float32))
        setattr(a, k, keep[k].ctypes.data_as(fp))
    pptv = "rho2" in co
    if pptv:      # mixing-ratio files as well (iout = 3)
        a.nx, a.ny, a.nz = (int(v) for v in co["met"])
        a.dx, a.dy, a.xlon0, a.ylat0 = (float(np.float32(v)) for v in co["metgeom"])
        a.dxout, a.dyout, a.outlon0, a.outlat0 = (float(np.float32(v)) for v in co["outgeom"][:4])
        for k, src in (("height", "height"), ("outheight", "outheight"), ("rho", "rho2"), ("weightmolar", "weightmolar")):
            keep[k] = np.ascontiguousarray(np.asarray(co[src], dtype=np.float32))
            setattr(a, k, keep[k].ctypes.data_as(fp))
    n3 = nxg * nyg * nzg
    wi = np.zeros(n3 + 1, np.int32); wr = np.zeros(n3 + 1, np.float32); g = np.zeros(n3, np.float32); f3 = np.zeros(n3, np.float32)
    buf = (C.c_ubyte * (64 + 3 * 16 + 3 * 8 * (n3 + 2)))()
    out = {}
    for ks in range(nspec):
        nb = lib.coo_concoutput(C.byref(a), ks, buf, wi.ctypes.data_as(C.POINTER(C.c_int32)), wr.ctypes.data_as(fp),
                                g.ctypes.data_as(fp), f3.ctypes.data_as(fp))
        out[f"_{ks + 1:03d}"] = bytes(buf[:nb])
        if pptv:
            lib.coo_pptvoutput.restype = C.c_long
            nb = lib.coo_pptvoutput(C.byref(a), ks, buf, wi.ctypes.data_as(C.POINTER(C.c_int32)), wr.ctypes.data_as(fp),
                                    g.ctypes.data_as(fp), f3.ctypes.data_as(fp))
            out[f"pptv_{ks + 1:03d}"] = bytes(buf[:nb])
    return out


# --------------------------------------------------------------------------
# releaseparticles + splitting (oracle/release_oracle.c)
# --------------------------------------------------------------------------
class _RloArgs(C.Structure):
    _dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    _fields_ = ([(k, C.c_int) for k in ("nx", "ny", "nz", "xglobal")]
                + [(k, C.c_double) for k in ("dx", "dy", "xlon0", "ylat0")]
                + [(k, C.POINTER(C.c_double)) for k in ("height", "oro", "rho2", "tt2")]
                + [(k, C.c_int) for k in ("nspec", "ldirect", "lsynctime", "mintime", "itsplit", "ind_rel", "mquasilag", "nclassunc", "ibdate", "ibtime")]
                + [("eps_nxmax", C.c_double), ("numpoint", C.c_int)]
                + [(k, C.POINTER(C.c_int)) for k in ("ireleasestart", "ireleaseend", "npart", "kindz")]
                + [(k, C.POINTER(C.c_double)) for k in ("xpoint1", "xpoint2", "ypoint1", "ypoint2", "zpoint1", "zpoint2", "xmass",
                                                         "point_hour", "area_hour", "point_dow", "area_dow")]
                + [("maxpart", C.c_long), ("numpart", C.c_long), ("numparticlecount", C.c_int)]
                + [(k, C.POINTER(C.c_double)) for k in ("xtra1", "ytra1", "ztra1", "uap", "xmass1")]
                + [(k, C.POINTER(C.c_int)) for k in ("itra1", "itramem", "itrasplit", "idt", "npoint", "nclass")]
                + [(k, C.POINTER(C.c_double)) for k in ("xmasssave", "rho_rel")]
                + [("ran1_idum", C.c_int), ("ran1_iy", C.c_int), ("ran1_iv", C.c_int * 32), ("status", C.c_int)]
                + [(k, C.c_int) for k in ("nest_on", "nxn", "nyn", "pad_")]
                + [(k, C.c_double) for k in ("xln", "yln", "xrn", "yrn", "xresoln", "yresoln")]
                + [(k, C.POINTER(C.c_double)) for k in ("oron", "rhon2", "ttn2")])


def rl_juldate(yyyymmdd, hhmiss, kind="r8"):
    build()
    lib = C.CDLL(os.path.join(HERE, f"librloracle_{kind}.so"))
    lib.rlo_juldate_pub.restype = C.c_double
    return lib.rlo_juldate_pub(int(yyyymmdd), int(hhmiss))


def rl_oracle(rs, kind="r8"):
    """The C restatement of releaseparticles + the splitting block on a synthetic.release_case() dict -> list of
    per-call dicts like scenario_io.run_rel_reference."""
    build()
    lib = C.CDLL(os.path.join(HERE, f"librloracle_{kind}.so"))
    a = _RloArgs()
    a.nx, a.ny, a.nz = (int(v) for v in rs["grid"])
    a.xglobal = int(rs["xglobal"])
    a.dx, a.dy, a.xlon0, a.ylat0 = (float(v) for v in rs["geom"])
    sw = [int(v) for v in rs["switches"]]
    a.nspec = int(rs["nspec"])
    a.ldirect, a.lsynctime, a.mintime, a.itsplit, a.ind_rel, a.mquasilag = sw[:6]
    maxpart, do_split = sw[6], sw[7]
    a.nclassunc = int(rs.get("nclassunc", 1))
    a.ibdate, a.ibtime = int(rs["bdate"][0]), int(rs["bdate"][1])
    a.numpoint = int(rs["numpoint"])
    keep = {}
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    for k in ("height", "oro", "rho2", "tt2", "xpoint1", "xpoint2", "ypoint1", "ypoint2", "zpoint1", "zpoint2", "xmass",
              "point_hour", "area_hour", "point_dow", "area_dow"):
        keep[k] = _f64(rs[k]); setattr(a, k, keep[k].ctypes.data_as(dp))
    for k, kk in (("ireleasestart", "ireleasestart"), ("ireleaseend", "ireleaseend"), ("npart", "npart_rel"), ("kindz", "kindz")):
        keep[k] = np.ascontiguousarray(np.asarray(rs[kk], dtype=np.int32)); setattr(a, k, keep[k].ctypes.data_as(ip))
    if "nest" in rs:
        a.nest_on = 1
        a.nxn, a.nyn = (int(v) for v in rs["nest"])
        a.xln, a.yln, a.xrn, a.yrn, a.xresoln, a.yresoln = (float(v) for v in rs["nestcorners"])
        a.eps_nxmax = float(rs["par_nxmax"])
        for k in ("oron", "rhon2", "ttn2"):
            keep[k] = _f64(rs[k]); setattr(a, k, keep[k].ctypes.data_as(dp))
    nsp = a.nspec
    P = {k: np.zeros(maxpart) for k in ("xtra1", "ytra1", "ztra1", "uap")}
    P["xmass1"] = np.zeros((nsp, maxpart))
    for k in ("itra1", "itramem", "itrasplit", "idt", "npoint", "nclass"):
        P[k] = np.zeros(maxpart, np.int32)
    P["itra1"][:] = -999999999; P["itrasplit"][:] = 999999999
    n0 = int(rs.get("npart", 0))
    rt = np.float32 if kind == "r4" else np.float64      # the host arrays hold default reals (xtra1, ytra1: always 8 bytes)
    for k in ("xtra1", "ytra1", "ztra1", "uap", "itra1", "itramem", "itrasplit", "idt", "npoint", "nclass"):
        if k in rs:
            P[k][:n0] = np.asarray(rs[k]).astype(rt) if k in ("ztra1", "uap") else np.asarray(rs[k])
    if "xmass1" in rs:
        P["xmass1"][:, :n0] = np.asarray(rs["xmass1"]).reshape(nsp, n0).astype(rt)
    for k, v in P.items():
        setattr(a, k, v.ctypes.data_as(dp if v.dtype == np.float64 else ip))
    xs, rr = np.zeros(a.numpoint), np.zeros(a.numpoint)
    a.xmasssave, a.rho_rel = xs.ctypes.data_as(dp), rr.ctypes.data_as(dp)
    a.maxpart, a.numpart, a.numparticlecount = maxpart, n0, 0
    lib.rlo_init(C.byref(a))
    calls = []
    for itime in (int(t) for t in rs["times"]):
        lib.rlo_releaseparticles(C.byref(a), itime)
        if a.status:
            raise RuntimeError("rlo_releaseparticles: no free storage space left")
        if do_split:
            lib.rlo_split(C.byref(a), itime)
        m = int(a.numpart)
        c = {k: v[:m].copy() for k, v in P.items() if k != "xmass1"}
        c["xmass1"] = P["xmass1"][:, :m].copy()
        c["state"] = np.array([itime, m, int(a.numparticlecount)], np.int32)
        c["xmasssave"], c["rho_rel"] = xs.copy(), rr.copy()
        calls.append(c)
    return calls


# --------------------------------------------------------------------------
# calcpar (oracle/calcpar_oracle.c)
# --------------------------------------------------------------------------
class _CpoArgs(C.Structure):
    _fields_ = ([(k, C.c_int) for k in ("nx", "ny", "nuvz", "lsubgrid")] + [("dy", C.c_double), ("ylat0", C.c_double)]
                + [(k, C.POINTER(C.c_double)) for k in ("ps", "tt2", "td2", "surfstr", "sshf", "excessoro", "tth", "qvh", "uuh", "vvh",
                                                         "akz", "bkz", "akm", "bkm", "ustar", "wstar", "oli", "hmix", "tropopause", "margin")])


def cp_oracle(m, cin, kind="r8"):
    """The C restatement of calcpar (ECMWF branch, without getvdep / calcpv) on a synthetic.model_levels() dict and
    synthetic.calcpar_inputs(): -> dict of ustar, wstar, oli, hmix, tropopause [ny][nx], and `margin`: the smallest relative
    distance of any of the column's level-search decisions from its threshold (test support)."""
    build()
    lib = C.CDLL(os.path.join(HERE, f"libcporacle_{kind}.so"))
    nx, ny, nz = (int(v) for v in m["grid"])
    a = _CpoArgs()
    a.nx, a.ny, a.nuvz, a.lsubgrid = nx, ny, nz, int(cin["lsubgrid"])
    a.dy, a.ylat0 = float(m["geom"][1]), float(m["geom"][3])
    keep = {}
    dp = C.POINTER(C.c_double)
    rt = np.float32 if kind == "r4" else np.float64
    for k, src in (("ps", m), ("tt2", m), ("td2", m), ("tth", m), ("qvh", m), ("uuh", m), ("vvh", m), ("akz", m), ("bkz", m),
                   ("surfstr", cin), ("sshf", cin), ("excessoro", cin), ("akm", cin), ("bkm", cin)):
        keep[k] = _f64(np.asarray(src[k]).astype(rt)); setattr(a, k, keep[k].ctypes.data_as(dp))
    out = {k: np.zeros((ny, nx)) for k in ("ustar", "wstar", "oli", "hmix", "tropopause", "margin")}
    for k, v in out.items():
        setattr(a, k, v.ctypes.data_as(dp))
    lib.cpo_calcpar(C.byref(a))
    return out


def cp_leaves(ps, t, td, stress, kind="r8"):
    """scalev(ps,t,td,stress), ew(td), f_qvsat(ps,t) of the restatement, element by element -> (n,3) float64."""
    build()
    lib = C.CDLL(os.path.join(HERE, f"libcporacle_{kind}.so"))
    ct = C.c_float if kind == "r4" else C.c_double
    for f in ("cpo_scalev", "cpo_ew", "cpo_f_qvsat"):
        getattr(lib, f).restype = ct
    lib.cpo_scalev.argtypes = [ct] * 4; lib.cpo_ew.argtypes = [ct]; lib.cpo_f_qvsat.argtypes = [ct] * 2
    rt = np.float32 if kind == "r4" else np.float64
    ps, t, td, stress = (np.asarray(v).astype(rt) for v in (ps, t, td, stress))
    o = np.empty((ps.size, 3))
    for i in range(ps.size):
        o[i] = (lib.cpo_scalev(ps[i], t[i], td[i], stress[i]), lib.cpo_ew(td[i]), lib.cpo_f_qvsat(ps[i], t[i]))
    return o


# ---------------------------------------------------------------------------------------------------------
# convective mixing (convmix / calcmatrix / convect43c / redist): oracle/convect_oracle.c
# ---------------------------------------------------------------------------------------------------------
class _CvoArgs(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nuvz", C.c_int), ("nconvlev", C.c_int), ("ldirect", C.c_int), ("lsynctime", C.c_int),
                ("itime", C.c_int), ("memtime1", C.c_int), ("memtime2", C.c_int),
                ("akz", C.c_void_p), ("bkz", C.c_void_p), ("akm", C.c_void_p), ("bkm", C.c_void_p),
                ("ps", C.c_void_p), ("tt2", C.c_void_p), ("td2", C.c_void_p), ("tth", C.c_void_p), ("qvh", C.c_void_p),
                ("height_nz", C.c_double), ("cbaseflux", C.c_void_p), ("numpart", C.c_long),
                ("xtra1", C.c_void_p), ("ytra1", C.c_void_p), ("ztra1", C.c_void_p), ("itra1", C.c_void_p),
                ("rn", C.c_void_p), ("lconv_col", C.c_void_p), ("nconvtop_col", C.c_void_p), ("fmassfrac_col", C.c_void_p),
                ("fm_col_id", C.c_void_p), ("fm_cap", C.c_int32), ("fm_count", C.c_int32), ("ran3_seeded", C.c_int32),
                ("state_words", C.c_int32 * 60), ("status", C.c_int32),
                ("nest_on", C.c_int32), ("nxn", C.c_int32), ("nyn", C.c_int32), ("pad_", C.c_int32),
                ("xln", C.c_double), ("yln", C.c_double), ("xrn", C.c_double), ("yrn", C.c_double), ("xresoln", C.c_double), ("yresoln", C.c_double),
                ("eps", C.c_double), ("psn", C.c_void_p), ("tt2n", C.c_void_p), ("td2n", C.c_void_p), ("tthn", C.c_void_p), ("qvhn", C.c_void_p),
                ("cbasefluxn", C.c_void_p)]


def conv_oracle(cs, kind="r8", fm_cap=8, ran3_words=None, between=None):
    """convmix on a synthetic.convection_case() dict, one entry per call: ztra1, cbaseflux, lconv / nconvtop per column,
    the random number each particle drew (-1: none), the redistribution matrices of the first fm_cap convective columns.
    ran3_words: the state of the module's ran3 stream to start from (Oracle.ran3_words(); redist's own seed, word 59, is
    set to its initial -88); default: an unseeded stream.  between(ic, arrays, words): called after every call with the
    particle arrays {x, y, z} and the stream state, all of which it may change in place (a test runs the trajectory step
    there, on the same stream, as timemanager does between two calls of convmix)."""
    build()
    lib = C.CDLL(os.path.join(HERE, f"libcvoracle_{kind}.so"))
    nx, ny, nuvz = (int(v) for v in cs["grid"])
    n = int(cs["npart"])
    nl = int(cs["nconvlev"])
    keep = {k: _f64(cs[k]) for k in ("akz", "bkz", "akm", "bkm", "ps", "tt2", "td2", "tth", "qvh", "xtra1", "ytra1")}
    if between is not None:                                  # the callback moves the particles: not in the caller's arrays
        keep["xtra1"], keep["ytra1"] = keep["xtra1"].copy(), keep["ytra1"].copy()
    z = _f64(cs["ztra1"]).copy()
    rt = np.float32 if kind == "r4" else np.float64
    z = z.astype(rt).astype(np.float64)                      # com_mod ztra1 is a default real
    cb = _f64(cs["cbaseflux"]).astype(rt).astype(np.float64)          # conv_mod cbaseflux likewise
    a = _CvoArgs()
    a.nx, a.ny, a.nuvz, a.nconvlev, a.ldirect, a.lsynctime = nx, ny, nuvz, nl, int(cs["ldirect"]), int(cs["lsynctime"])
    a.memtime1, a.memtime2 = int(cs["memtime"][0]), int(cs["memtime"][1])
    for k in ("akz", "bkz", "akm", "bkm", "ps", "tt2", "td2", "tth", "qvh", "xtra1", "ytra1"):
        setattr(a, k, keep[k].ctypes.data)
    a.height_nz = float(cs["height_nz"])
    a.cbaseflux = cb.ctypes.data
    a.numpart = n
    a.ztra1 = z.ctypes.data
    a.fm_cap = fm_cap
    a.ran3_seeded = 0
    if ran3_words is not None:
        a.ran3_seeded = 1
        for i in range(59):
            a.state_words[i] = int(ran3_words[i])
        a.state_words[59] = -88
    cbn = None
    if "nest" in cs:
        a.nest_on = 1
        a.nxn, a.nyn = (int(v) for v in cs["nest"])
        a.xln, a.yln, a.xrn, a.yrn, a.xresoln, a.yresoln = (float(rt(v)) for v in cs["nestgeom"])
        a.eps = float(rt(rt(cs["par_nxmax"]) / rt(3.e5)))
        for k in ("psn", "tt2n", "td2n", "tthn", "qvhn"):
            keep[k] = _f64(cs[k])
            setattr(a, k, keep[k].ctypes.data)
        cbn = _f64(cs["cbasefluxn"]).astype(rt).astype(np.float64)
        a.cbasefluxn = cbn.ctypes.data
    out = []
    for ic, itime in enumerate(int(t) for t in cs["itimes"]):
        itra1 = np.where(np.asarray(cs["due"])[:, ic], itime, itime + 12345).astype(np.int32)
        rn = np.zeros(n)
        lc = np.zeros((ny, nx), np.int32); nt = np.zeros((ny, nx), np.int32)
        fm = np.zeros((fm_cap, nl, nl)); fid = np.full(fm_cap, -1, np.int32)
        a.itime = itime
        a.itra1 = itra1.ctypes.data
        a.rn, a.lconv_col, a.nconvtop_col = rn.ctypes.data, lc.ctypes.data, nt.ctypes.data
        a.fmassfrac_col, a.fm_col_id = fm.ctypes.data, fid.ctypes.data
        lib.cvo_convmix(C.byref(a))
        out.append(dict(ztra1=z.copy(), cbaseflux=cb.copy(), lconv=lc, nconvtop=nt, rn=rn, fmassfrac=fm, fm_col=fid, fm_count=int(a.fm_count)))
        if cbn is not None:
            out[-1]["cbasefluxn"] = cbn.copy()
        if between is not None:
            words = np.array(a.state_words[:], np.int32)
            between(ic, dict(x=keep["xtra1"], y=keep["ytra1"], z=z), words)
            for i in range(60):
                a.state_words[i] = int(words[i])
    return out
