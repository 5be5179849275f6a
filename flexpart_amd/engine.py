"""Host-side mirror of the reference's particle loop, on top of the C ABI.

`Engine` plays the role of the Fortran host around the replaced block
(reference src/timemanager.f90:531-712): it owns com_mod-shaped host arrays,
hands them to the GPU engine through include/flexpart_amd.h, and exposes the
same vocabulary (xtra1, ytra1, ztra1, uap, ucp, uzp, us, vs, ws, idt, itra1,
itramem, cbt, xmass1; memtime/memind; lsynctime ...).  All arithmetic happens
in the HIP library; nothing here computes trajectories.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import (FpxConfig, FpxFields, FpxNests, FpxOutgrid, FpxOutgridNest, FpxParticles, FpxStepStats, FpxWetConfig, FpxWetFields, RNG_PHILOX,
                   RNG_TABLE_COUNTER, RNG_TABLE_SEQ, check)

# polar stereographic set-up is host work in the reference (gridcheck_ecmwf.f90:341-366 via
# cmapf_mod stlmbr/stcm2p); the engine only consumes the resulting 9-number map records.
SWITCHNORTH, SWITCHSOUTH = 75.0, -75.0


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def release_tables(sc):
    """(xmass [nspec][numpoint], npart [numpoint]) of a scenario: keys numpoint, xmass, npart_rel; the default is
    the single release point of unit mass holding every particle."""
    nspec = int(sc["nspec"])
    numpoint = int(sc.get("numpoint", 1))
    xm = np.asarray(sc.get("xmass", np.ones(nspec * numpoint)), dtype=np.float64).reshape(nspec, numpoint)
    npt = np.asarray(sc.get("npart_rel", np.full(numpoint, max(int(sc.get("npart", 0)), 1))), dtype=np.int32).ravel()
    return xm, npt


class Engine:
    def __init__(self, sc, *, compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_TABLE_SEQ,
                 seed=0x5EED, max_particles=None, device=0, pad=(0, 0, 0), sort_interval=0,
                 polemaps=None, particle_base=0, blend_mode=0, global_particles=0, pbl_slice_passes=0, options=None):
        """sc: scenario dict (flexpart_amd.synthetic).  pad: extra allocated (nxmax-nx,
        nymax-ny, nzmax-nz) to exercise the reference's padded-array convention."""
        self.lib = _lib.load()
        self.sc = sc
        self.hreal = np.float32 if host_real_bytes == 4 else np.float64
        nx, ny, nz = (int(v) for v in sc["grid"])
        self.nx, self.ny, self.nz = nx, ny, nz
        self.nxmax, self.nymax, self.nzmax = nx + pad[0], ny + pad[1], nz + pad[2]
        dx, dy, xlon0, ylat0 = (float(v) for v in sc["geom"])
        xg, ng, sg = (int(v) for v in sc["globalflags"])
        nspec = int(sc["nspec"])
        self.nspec = nspec
        n = int(sc.get("npart", 0))
        cfg = FpxConfig()
        cfg.struct_bytes = C.sizeof(FpxConfig)
        cfg.device = device
        cfg.compute_real_bytes = compute_real_bytes
        cfg.host_real_bytes = host_real_bytes
        cfg.max_particles = int(max_particles or max(n, 1))
        cfg.nx, cfg.ny, cfg.nz, cfg.nmixz = nx, ny, nz, int(sc.get("nmixz", nz))   # without height/nmixz: set by verttransform(init)
        cfg.nxmax, cfg.nymax, cfg.nzmax = self.nxmax, self.nymax, self.nzmax
        cfg.dx, cfg.dy, cfg.xlon0, cfg.ylat0 = dx, dy, xlon0, ylat0
        cfg.xglobal, cfg.nglobal, cfg.sglobal = xg, ng, sg
        rt = self.hreal
        # (switchnorth-ylat0)/dy in the host's default real kind, gridcheck_ecmwf.f90:348,362
        cfg.switchnorthg = float((rt(SWITCHNORTH) - rt(ylat0)) / rt(dy)) if ng else 999999.0
        cfg.switchsouthg = float((rt(SWITCHSOUTH) - rt(ylat0)) / rt(dy)) if sg else 999999.0
        if ng or sg:
            if polemaps is None:   # the host has no com_mod records: build them as gridcheck does
                north = (C.c_double * 9)()
                south = (C.c_double * 9)()
                check(self.lib.fpx_polar_maps(host_real_bytes, dy, north, south), "fpx_polar_maps")
                polemaps = (list(north), list(south))
            for i in range(9):
                cfg.northpolemap[i] = float(polemaps[0][i])
                cfg.southpolemap[i] = float(polemaps[1][i])
        for k in ("ldirect", "lsynctime", "method", "mintime", "ifine", "turbswitch", "cblflag",
                  "mdomainfill", "lsettling"):
            setattr(cfg, k, int(sc[k]))
        cfg.ctl = float(sc["ctl"])
        cfg.d_trop, cfg.d_strat, cfg.turbmesoscale = (float(v) for v in sc["turbpar"])
        cfg.nspec = nspec
        cfg.maxspec = nspec
        cfg.drydep = int(sc["drydep"])
        for i in range(nspec):
            cfg.drydepspec[i] = int(np.asarray(sc["drydepspec"]).ravel()[i])
            cfg.density[i] = float(sc["density"][i]); cfg.dquer[i] = float(sc["dquer"][i])
            cfg.vsetaver[i] = float(sc["vsetaver"][i]); cfg.cunningham[i] = float(sc["cunningham"][i])
            cfg.decay[i] = float(sc["decay"][i])
        cfg.mquasilag = int(sc.get("mquasilag", 0))
        cfg.lage_last = int(np.asarray(sc["lage"]).ravel()[-1])
        cfg.rng_mode = rng_mode
        cfg.seed = seed
        cfg.sort_interval = sort_interval
        # eps = nxmax/3.e5 uses the par_mod nxmax of the host build (advance.f90:107); scenarios carry
        # it so that runs compare with the reference binary they were pinned against (361 or 721)
        cfg.par_nxmax = int(sc.get("par_nxmax", 361))
        cfg.particle_base = int(particle_base or sc.get("particle_base", 0))
        cfg.drybkdep, cfg.wetbkdep = int(sc.get("drybkdep", 0)), int(sc.get("wetbkdep", 0))   # backward runs with receptor scavenging
        self.bkdep = bool(cfg.drybkdep or cfg.wetbkdep)
        cfg.turboff, cfg.interpolhmix = int(sc.get("turboff", 0)), int(sc.get("interpolhmix", 0))   # com_mod.f90:777-778
        cfg.ipout, cfg.iflux, cfg.linit_cond = int(sc.get("ipout", 0)), int(sc.get("iflux", 0)), int(sc.get("linit_cond", 0))   # refused when set
        cfg.blend_mode = int(blend_mode)                # 0: from global_particles, 1 on, 2 off
        cfg.global_particles = int(global_particles)    # the run's particle count over all ranks
        cfg.pbl_slice_passes = int(pbl_slice_passes)    # 0: the engine's schedule, -1: one launch, k: k passes per launch
        self.cfg = cfg
        self.h = C.c_void_p()
        check(self.lib.fpx_create(C.byref(self.h), C.byref(cfg)), "fpx_create")
        for name, value in (options or {}).items():
            self.set_option(name, value)
        if "height" in sc:
            hgt = np.ascontiguousarray(np.asarray(sc["height"]).astype(rt))
            check(self.lib.fpx_set_height(self.h, _vp(hgt), nz), "fpx_set_height")
        self.itime = int(sc.get("itime0", 0))
        self.lsynctime = int(sc["lsynctime"])
        self.n = 0
        self.set_release_points(*release_tables(sc))
        if "zpoint1" in sc and "zpoint2" in sc:      # point_mod zpoint1/zpoint2: WETBKDEP multiplies with the release's height range
            z1 = np.ascontiguousarray(np.asarray(sc["zpoint1"]).astype(rt))
            z2 = np.ascontiguousarray(np.asarray(sc["zpoint2"]).astype(rt))
            check(self.lib.fpx_set_release_heights(self.h, z1.size, _vp(z1), _vp(z2)), "fpx_set_release_heights")
        if rng_mode != RNG_PHILOX:
            check(self.lib.fpx_rng_fill_table(self.h), "fpx_rng_fill_table")
        if "uu" in sc:
            self.upload_fields_from_scenario(sc)
        if "nest" in sc:
            self.upload_nests_from_scenario(sc)
        if n:
            self.upload_particles_from_scenario(sc)
        self.gshape = None
        if "outgrid" in sc:
            self.outgrid_from_scenario(sc)
        self.has_wet = bool(sc.get("wetdep", 0))
        self.has_conv = False                        # set by conv_init: run() then calls convmix where timemanager does
        if self.has_wet:
            self.wet_from_scenario(sc)

    # ---- met fields ---------------------------------------------------------
    def _host3(self, a, m):
        """slot m of a compact (2,nz,ny,nx) array -> com_mod-shaped padded host array."""
        out = np.zeros((self.nzmax, self.nymax, self.nxmax), self.hreal)
        out[: self.nz, : self.ny, : self.nx] = a[m]
        return out

    def _host2(self, a, m):
        out = np.zeros((self.nymax, self.nxmax), self.hreal)
        out[: self.ny, : self.nx] = a[m]
        return out

    def upload_fields_from_scenario(self, sc):
        for m in (0, 1):
            keep = {}
            f = FpxFields()
            for k in ("uu", "vv", "ww", "uupol", "vvpol", "rho", "drhodz", "tt"):
                if k in sc:
                    keep[k] = self._host3(sc[k], m)
                    setattr(f, k, keep[k].ctypes.data)
            for k in ("hmix", "ustar", "wstar", "oli", "tropopause"):
                keep[k] = self._host2(sc[k], m)
                setattr(f, k, keep[k].ctypes.data)
            if "vdep" in sc:
                v = np.zeros((self.nspec, self.nymax, self.nxmax), self.hreal)
                v[:, : self.ny, : self.nx] = sc["vdep"][m]
                keep["vdep"] = v
                f.vdep = v.ctypes.data
            check(self.lib.fpx_upload_fields(self.h, m + 1, C.byref(f)), "fpx_upload_fields")
        self.set_windtime(sc["memtime"], sc["memind"])

    def verttransform(self, slot, m, sfc, *, init=False, want=("uu", "vv", "ww", "tt", "qv", "pv", "rho", "drhodz", "uupol", "vvpol"),
                      host_arrays=None, nest=None):
        """fpx_verttransform_ecmwf: m = synthetic.model_levels() dict (compact [nz][ny][nx] arrays),
        sfc = dict of compact 2-D fields (hmix, ustar, wstar, oli, tropopause[, vdep]) for this slot.
        Returns the z-level arrays asked for (compact), height and nmixz."""
        from ._lib import FpxModelLevels, FpxFieldsOut
        rt = self.hreal
        if nest is not None:      # m is the nest's input (synthetic.nest_model_levels): fpx_verttransform_nest, nest 1
            return self._verttransform_nest(slot, m, sfc, want)
        keep = {} if host_arrays is None else host_arrays     # host_arrays: a dict the caller keeps alive -> arrays stay put and are pinned
        ml = FpxModelLevels()
        ml.pin_host = 0 if host_arrays is None else 1
        for k in ("uuh", "vvh", "pvh", "wwh", "tth", "qvh"):
            if k not in keep:
                keep[k] = np.zeros((self.nzmax, self.nymax, self.nxmax), rt)
            keep[k][: self.nz, : self.ny, : self.nx] = m[k]
            setattr(ml, k, keep[k].ctypes.data)
        for k in ("ps", "tt2", "td2"):
            if k not in keep:
                keep[k] = np.zeros((self.nymax, self.nxmax), rt)
            keep[k][: self.ny, : self.nx] = m[k]
            setattr(ml, k, keep[k].ctypes.data)
        for k in ("akz", "bkz", "aknew", "bknew"):
            keep[k] = np.ascontiguousarray(np.asarray(m[k]).astype(rt))
            setattr(ml, k, keep[k].ctypes.data)
        ml.nuvz = ml.nwz = self.nz
        ml.init = int(init)
        f = FpxFields()
        for k in ("hmix", "ustar", "wstar", "oli", "tropopause"):
            if sfc is None:
                break
            a = np.zeros((self.nymax, self.nxmax), rt)
            a[: self.ny, : self.nx] = sfc[k]
            keep["s" + k] = a
            setattr(f, k, a.ctypes.data)
        if sfc is not None and "vdep" in sfc:
            v = np.zeros((self.nspec, self.nymax, self.nxmax), rt)
            v[:, : self.ny, : self.nx] = sfc["vdep"]
            keep["svdep"] = v
            f.vdep = v.ctypes.data
        o = FpxFieldsOut()
        res = {}
        for k in want:
            res[k] = np.zeros((self.nzmax, self.nymax, self.nxmax), rt)
            setattr(o, k, res[k].ctypes.data)
        hgt = np.zeros(self.nz, rt)
        o.height = hgt.ctypes.data
        nmixz = C.c_int32(0)
        o.nmixz = C.pointer(nmixz)
        import time as _time
        t0 = _time.perf_counter()
        check(self.lib.fpx_verttransform_ecmwf(self.h, int(slot), C.byref(ml), C.byref(f) if sfc is not None else None, C.byref(o)),
              "fpx_verttransform_ecmwf")
        call_s = _time.perf_counter() - t0          # the C call alone (the marshalling above is the mirror's, not the host's)
        out = {k: v[: self.nz, : self.ny, : self.nx].astype(np.float64) for k, v in res.items()}
        out["height"] = hgt.astype(np.float64)
        out["nmixz"] = int(nmixz.value)
        ms = C.c_double(0)
        check(self.lib.fpx_verttransform_time(self.h, C.byref(ms)), "fpx_verttransform_time")
        out["device_ms"] = ms.value
        out["call_ms"] = call_s * 1e3
        return out

    def calcpar(self, slot, cin, vdep=None):
        """fpx_calcpar after verttransform(slot, m, None): cin = synthetic.calcpar_inputs(m).  Returns the five 2-D fields
        (compact [ny][nx]) and the device time."""
        from ._lib import FpxCalcparIn, FpxCalcparOut
        rt = self.hreal
        keep = {}
        c = FpxCalcparIn()
        for k in ("surfstr", "sshf", "excessoro"):
            a = np.zeros((self.nymax, self.nxmax), rt)
            a[: self.ny, : self.nx] = cin[k]
            keep[k] = a
            setattr(c, k, a.ctypes.data)
        for k in ("akm", "bkm"):
            keep[k] = np.ascontiguousarray(np.asarray(cin[k]).astype(rt))
            setattr(c, k, keep[k].ctypes.data)
        c.lsubgrid = int(cin["lsubgrid"])
        if vdep is not None:
            v = np.zeros((self.nspec, self.nymax, self.nxmax), rt)
            v[:, : self.ny, : self.nx] = vdep
            keep["vdep"] = v
            c.vdep = v.ctypes.data
        o = FpxCalcparOut()
        res = {}
        for k in ("ustar", "wstar", "oli", "hmix", "tropopause"):
            res[k] = np.zeros((self.nymax, self.nxmax), rt)
            setattr(o, k, res[k].ctypes.data)
        check(self.lib.fpx_calcpar(self.h, int(slot), C.byref(c), C.byref(o)), "fpx_calcpar")
        out = {k: v[: self.ny, : self.nx].astype(np.float64) for k, v in res.items()}
        ms = C.c_double(0)
        check(self.lib.fpx_calcpar_time(self.h, C.byref(ms)), "fpx_calcpar_time")
        out["device_ms"] = ms.value
        return out

    def _verttransform_nest(self, slot, n, sfc, want):
        from ._lib import FpxModelLevels, FpxFieldsOut
        rt = self.hreal
        nxn, nyn = int(n["grid"][0]), int(n["grid"][1])
        keep = {}
        ml = FpxModelLevels()
        for k in ("uuh", "vvh", "pvh", "wwh", "tth", "qvh"):
            keep[k] = np.ascontiguousarray(np.zeros((self.nzmax, nyn, nxn), rt))
            keep[k][: self.nz] = n[k]
            setattr(ml, k, keep[k].ctypes.data)
        for k in ("ps", "tt2", "td2"):
            keep[k] = np.ascontiguousarray(np.asarray(n[k]).astype(rt))
            setattr(ml, k, keep[k].ctypes.data)
        for k in ("akz", "bkz", "aknew", "bknew"):
            keep[k] = np.ascontiguousarray(np.asarray(n[k]).astype(rt))
            setattr(ml, k, keep[k].ctypes.data)
        ml.nuvz = ml.nwz = self.nz
        ml.nest_dy, ml.nest_ylat0 = float(rt(n["geom"][1])), float(rt(n["geom"][3]))
        f = FpxFields()
        for k in ("hmix", "ustar", "wstar", "oli", "tropopause"):
            keep["s" + k] = np.ascontiguousarray(np.asarray(sfc[k]).astype(rt))
            setattr(f, k, keep["s" + k].ctypes.data)
        o = FpxFieldsOut()
        res = {}
        for k in want:
            if k in ("uupol", "vvpol"):
                continue
            res[k] = np.zeros((self.nzmax, nyn, nxn), rt)
            setattr(o, k, res[k].ctypes.data)
        check(self.lib.fpx_verttransform_nest(self.h, 1, int(slot), C.byref(ml), C.byref(f), C.byref(o)), "fpx_verttransform_nest")
        return {k: v[: self.nz].astype(np.float64) for k, v in res.items()}

    def upload_diag_fields_from_scenario(self, sc):
        """oro and, for both slots, pv, qv, tt (compact arrays) -> fpx_upload_diag_fields."""
        from ._lib import FpxDiagFields
        rt = self.hreal
        oro = np.zeros((self.nymax, self.nxmax), rt)
        oro[: self.ny, : self.nx] = sc["oro"]
        f = FpxDiagFields()
        f.oro = oro.ctypes.data
        check(self.lib.fpx_upload_diag_fields(self.h, 0, C.byref(f)), "fpx_upload_diag_fields")
        for m in (0, 1):
            keep = {}
            f = FpxDiagFields()
            for k in ("pv", "qv", "tt"):
                keep[k] = self._host3(sc[k], m)
                setattr(f, k, keep[k].ctypes.data)
            check(self.lib.fpx_upload_diag_fields(self.h, m + 1, C.byref(f)), "fpx_upload_diag_fields")

    def upload_diag_nest_fields(self, nest, oron, ttn2=None):
        """oron [nyn][nxn] and ttn of time slot 2 [nz][nyn][nxn] of a nested wind field (compact nests in this mirror)."""
        from ._lib import FpxDiagFields
        rt = self.hreal
        f = FpxDiagFields()
        o = np.ascontiguousarray(np.asarray(oron).astype(rt))
        f.oro = o.ctypes.data
        t = None
        if ttn2 is not None:
            t = np.zeros((self.nzmax,) + tuple(np.asarray(ttn2).shape[1:]), rt)
            t[: np.asarray(ttn2).shape[0]] = ttn2
            f.tt = t.ctypes.data
        check(self.lib.fpx_upload_diag_nest_fields(self.h, int(nest), 2 if ttn2 is not None else 0, C.byref(f)), "fpx_upload_diag_nest_fields")

    def partoutput(self, itime, path):
        """fpx_partoutput: writes the reference's partposit_* dump to `path`; returns the number of particle records."""
        n = C.c_int64(0)
        check(self.lib.fpx_partoutput(self.h, int(itime), str(path).encode(), C.byref(n)), "fpx_partoutput")
        ms = C.c_double(0)
        check(self.lib.fpx_partoutput_time(self.h, C.byref(ms)), "fpx_partoutput_time")
        self.partoutput_device_ms = ms.value
        return int(n.value)

    def readpartpositions(self, path, jul_header, bdate, mintime, nclassunc=1, itrasplit=999999999):
        """fpx_readpartpositions: warm start from the dump `path`; -> (numpart, numparticlecount, itimein)."""
        from ._lib import FpxRestart
        r = FpxRestart(float(jul_header), float(bdate), int(mintime), int(nclassunc), int(itrasplit), 0)
        n = C.c_int64(0); npc = C.c_int32(0); it = C.c_int32(0)
        check(self.lib.fpx_readpartpositions(self.h, str(path).encode(), C.byref(r), C.byref(n), C.byref(npc), C.byref(it)),
              "fpx_readpartpositions")
        self.n = int(n.value)
        return int(n.value), int(npc.value), int(it.value)

    # ---- convective mixing (fpx_conv_init / fpx_upload_conv_fields / fpx_convmix) ----
    def conv_init(self, cs):
        """Level structure of a synthetic.convection_case()-like dict: grid[2] = nuvz, nconvlev, akz, bkz, akm, bkm."""
        from ._lib import FpxConvConfig
        rt = self.hreal
        c = FpxConvConfig()
        c.struct_bytes = C.sizeof(FpxConvConfig)
        c.nuvz, c.nconvlev = int(cs["grid"][2]), int(cs["nconvlev"])
        keep = {k: np.ascontiguousarray(np.asarray(cs[k]).astype(rt)) for k in ("akz", "bkz", "akm", "bkm")}
        for k, a in keep.items():
            setattr(c, k, a.ctypes.data)
        check(self.lib.fpx_conv_init(self.h, C.byref(c)), "fpx_conv_init")
        self.conv_nuvz = c.nuvz
        self.has_conv = True

    def upload_conv_fields(self, slot, ps, tt2, td2, tth, qvh, nuvzmax=None):
        """One wind-field slot of ps, tt2, td2 [ny][nx] and tth, qvh [nuvz][ny][nx] (compact), padded to the host layout."""
        from ._lib import FpxConvFields
        rt = self.hreal
        nuvz = int(np.asarray(tth).shape[0])
        nuvzmax = nuvz + 1 if nuvzmax is None else int(nuvzmax)
        f = FpxConvFields()
        keep = {}
        for k, a in (("ps", ps), ("tt2", tt2), ("td2", td2)):
            b = np.zeros((self.nymax, self.nxmax), rt)
            b[: self.ny, : self.nx] = a
            keep[k] = b
            setattr(f, k, b.ctypes.data)
        for k, a in (("tth", tth), ("qvh", qvh)):
            b = np.zeros((nuvzmax, self.nymax, self.nxmax), rt)
            b[:nuvz, : self.ny, : self.nx] = a
            keep[k] = b
            setattr(f, k, b.ctypes.data)
        f.nuvzmax = nuvzmax
        check(self.lib.fpx_upload_conv_fields(self.h, int(slot), C.byref(f)), "fpx_upload_conv_fields")

    def upload_conv_nest_fields(self, nest, slot, ps, tt2, td2, tth, qvh, nuvzmax=None):
        """The same five arrays of nested wind field `nest` [nyn][nxn] / [nuvz][nyn][nxn] (this mirror declares compact nests:
        nxmaxn = nxn, nymaxn = nyn)."""
        from ._lib import FpxConvFields
        rt = self.hreal
        nuvz, nyn, nxn = (int(v) for v in np.asarray(tth).shape)
        nuvzmax = nuvz + 1 if nuvzmax is None else int(nuvzmax)
        f = FpxConvFields()
        keep = {}
        for k, a in (("ps", ps), ("tt2", tt2), ("td2", td2)):
            b = np.ascontiguousarray(np.asarray(a).astype(rt))
            keep[k] = b
            setattr(f, k, b.ctypes.data)
        for k, a in (("tth", tth), ("qvh", qvh)):
            b = np.zeros((nuvzmax, nyn, nxn), rt)
            b[:nuvz] = a
            keep[k] = b
            setattr(f, k, b.ctypes.data)
        f.nuvzmax = nuvzmax
        check(self.lib.fpx_upload_conv_nest_fields(self.h, int(nest), int(slot), C.byref(f)), "fpx_upload_conv_nest_fields")

    def cbaseflux_nest(self, nest, shape, new=None):
        a = np.zeros(shape, self.hreal)
        if new is not None:
            a[:] = new
            check(self.lib.fpx_set_cbaseflux_nest(self.h, int(nest), a.ctypes.data), "fpx_set_cbaseflux_nest")
            return None
        check(self.lib.fpx_get_cbaseflux_nest(self.h, int(nest), a.ctypes.data), "fpx_get_cbaseflux_nest")
        return a.astype(np.float64)

    def convmix(self, itime=None):
        """fpx_convmix: -> number of particles whose height was set; .convmix_device_ms holds the device time."""
        n = C.c_int64(0)
        check(self.lib.fpx_convmix(self.h, int(self.itime if itime is None else itime), C.byref(n)), "fpx_convmix")
        ms = C.c_double(0)
        check(self.lib.fpx_convmix_time(self.h, C.byref(ms)), "fpx_convmix_time")
        self.convmix_device_ms = ms.value
        return int(n.value)

    def cbaseflux(self, new=None):
        a = np.zeros((self.ny, self.nx), self.hreal)
        if new is not None:
            a[:] = new
            check(self.lib.fpx_set_cbaseflux(self.h, a.ctypes.data), "fpx_set_cbaseflux")
            return None
        check(self.lib.fpx_get_cbaseflux(self.h, a.ctypes.data), "fpx_get_cbaseflux")
        return a.astype(np.float64)

    def checkpoint_write(self, path, itime=None, numparticlecount=None):
        """fpx_checkpoint_write: everything the particle loop carries (lossless, unlike partoutput)."""
        check(self.lib.fpx_checkpoint_write(self.h, str(path).encode(), int(self.itime if itime is None else itime),
                                            int(getattr(self, "numparticlecount", 0) if numparticlecount is None else numparticlecount)),
              "fpx_checkpoint_write")

    def checkpoint_read(self, path):
        """fpx_checkpoint_read: -> (itime, numpart, numparticlecount); the engine continues from there."""
        it = C.c_int32(0); n = C.c_int64(0); npc = C.c_int32(0)
        check(self.lib.fpx_checkpoint_read(self.h, str(path).encode(), C.byref(it), C.byref(n), C.byref(npc)), "fpx_checkpoint_read")
        self.n = int(n.value)
        self.itime = int(it.value)
        self.numparticlecount = int(npc.value)
        return self.itime, self.n, self.numparticlecount

    def concoutput(self, itime, prefix, area, volume, outnum, wetdep=False, drydep=False, clear=False, nest=False,
                   iout=1, prefix_pptv=None, outheight=None, outlon0=0.0, outlat0=0.0, weightmolar=(), reduced=False):
        """fpx_concoutput: writes <prefix><nnn> (the reference's grid_conc_* files) for every species."""
        from ._lib import FpxConcout
        a = np.ascontiguousarray(np.asarray(area, dtype=np.float32))
        v = np.ascontiguousarray(np.asarray(volume, dtype=np.float32))
        c = FpxConcout(a.ctypes.data, v.ctypes.data, float(outnum), int(wetdep), int(drydep), int(nest), int(iout))
        if prefix_pptv is not None:
            oh = np.ascontiguousarray(np.asarray(outheight, dtype=np.float32))
            c.prefix_pptv = str(prefix_pptv).encode(); c.outheight = oh.ctypes.data
            c.outlon0, c.outlat0 = float(outlon0), float(outlat0)
            for i, w in enumerate(weightmolar):
                c.weightmolar[i] = float(w)
        c.reduced = int(reduced)
        check(self.lib.fpx_concoutput(self.h, int(itime), C.byref(c), str(prefix).encode(), int(clear)), "fpx_concoutput")

    def upload_nests_from_scenario(self, sc):
        """One nested grid: geometry as gridcheck_nests.f90:362-378 derives it, fields uun, vvn, ..."""
        rt = self.hreal
        self.init_nest(sc["nest"], sc["nestgeom"])
        for m in (0, 1):
            self._upload_nest_slot(sc, m)

    def init_nest(self, nest, nestgeom):
        """fpx_nests_init for one nested grid: geometry as gridcheck_nests.f90:359-372 derives it."""
        rt = self.hreal
        nxn, nyn = (int(v) for v in nest)
        dxn, dyn, xlon0n, ylat0n = (rt(v) for v in nestgeom)
        dx, dy, xlon0, ylat0 = (rt(self.cfg.dx), rt(self.cfg.dy), rt(self.cfg.xlon0), rt(self.cfg.ylat0))
        n = FpxNests()
        n.struct_bytes = C.sizeof(FpxNests)
        n.numbnests = 1
        n.nxmaxn, n.nymaxn = nxn, nyn
        n.nxn[0], n.nyn[0] = nxn, nyn
        xaux2 = xlon0n + rt(nxn - 1) * dxn
        yaux2 = ylat0n + rt(nyn - 1) * dyn
        n.xresoln[0] = float(dx / dxn); n.yresoln[0] = float(dy / dyn)
        n.xln[0] = float((xlon0n - xlon0) / dx); n.xrn[0] = float((xaux2 - xlon0) / dx)
        n.yln[0] = float((ylat0n - ylat0) / dy); n.yrn[0] = float((yaux2 - ylat0) / dy)
        check(self.lib.fpx_nests_init(self.h, C.byref(n)), "fpx_nests_init")

    def _upload_nest_slot(self, sc, m):
        rt = self.hreal
        if True:
            keep = {}
            f = FpxFields()
            for k, kn in (("uu", "uun"), ("vv", "vvn"), ("ww", "wwn"), ("rho", "rhon"), ("drhodz", "drhodzn")):
                keep[k] = np.ascontiguousarray(np.asarray(sc[kn])[m].astype(rt))
                setattr(f, k, keep[k].ctypes.data)
            for k, kn in (("hmix", "hmixn"), ("ustar", "ustarn"), ("wstar", "wstarn"), ("oli", "olin"),
                          ("tropopause", "tropopausen")):
                keep[k] = np.ascontiguousarray(np.asarray(sc[kn])[m].astype(rt))
                setattr(f, k, keep[k].ctypes.data)
            if "vdepn" in sc:
                keep["vdep"] = np.ascontiguousarray(np.asarray(sc["vdepn"])[m].astype(rt))
                f.vdep = keep["vdep"].ctypes.data
            check(self.lib.fpx_upload_nest_fields(self.h, 1, m + 1, C.byref(f)), "fpx_upload_nest_fields")

    def set_release_points(self, xmass, npart):
        """point_mod xmass(numpoint,maxspec) (given as [nspec][numpoint]) and npart(numpoint)."""
        xm = np.ascontiguousarray(np.asarray(xmass, dtype=np.float64).reshape(self.nspec, -1).astype(self.hreal))
        npt = np.ascontiguousarray(np.asarray(npart, dtype=np.int32).ravel())
        assert xm.shape[1] == npt.size
        check(self.lib.fpx_set_release_points(self.h, int(npt.size), _vp(xm), npt.ctypes.data_as(C.POINTER(C.c_int32))),
              "fpx_set_release_points")
        self.numpoint = int(npt.size)

    # ---- releaseparticles + splitting on the device ------------------------------------------------------
    def release_init(self, rs):
        """fpx_release_init from a release scenario dict (synthetic.release_case): the point_mod / com_mod tables."""
        from ._lib import FpxRelease
        rt = self.hreal
        nsp, np_ = self.nspec, int(rs["numpoint"])
        self.set_release_points(np.asarray(rs["xmass"]).reshape(nsp, np_), rs["npart_rel"])
        keep = {}
        r = FpxRelease()
        r.struct_bytes = C.sizeof(FpxRelease)
        r.numpoint = np_
        for k in ("ireleasestart", "ireleaseend"):
            keep[k] = np.ascontiguousarray(np.asarray(rs[k], dtype=np.int32)); setattr(r, k, keep[k].ctypes.data)
        keep["kindz"] = np.ascontiguousarray(np.asarray(rs["kindz"], dtype=np.int16)); r.kindz = keep["kindz"].ctypes.data
        for k in ("xpoint1", "xpoint2", "ypoint1", "ypoint2", "zpoint1", "zpoint2"):
            keep[k] = np.ascontiguousarray(np.asarray(rs[k]).astype(rt)); setattr(r, k, keep[k].ctypes.data)
        for k, n2 in (("point_hour", 24), ("area_hour", 24), ("point_dow", 7), ("area_dow", 7)):
            if k in rs:       # given compact [n2][nspec]; the host arrays are (maxspec, n2) column-major with maxspec = nspec here
                keep[k] = np.ascontiguousarray(np.asarray(rs[k]).reshape(n2, nsp).astype(rt)); setattr(r, k, keep[k].ctypes.data)
        r.bdate = float(rs["bdate_jul"])
        sw = [int(v) for v in rs["switches"]]
        r.itsplit, r.ind_rel, r.nclassunc = sw[3], sw[4], int(rs.get("nclassunc", 1))
        check(self.lib.fpx_release_init(self.h, C.byref(r)), "fpx_release_init")
        self.xmasssave = np.zeros(np_, rt)
        self.rho_rel = np.zeros(np_, rt)
        self.numparticlecount = 0

    def releaseparticles(self, itime):
        """fpx_releaseparticles; returns the number of particles released; self.n (numpart) follows."""
        n = C.c_int64(self.n); npc = C.c_int32(self.numparticlecount); nrel = C.c_int64(0)
        check(self.lib.fpx_releaseparticles(self.h, int(itime), C.byref(n), C.byref(npc), _vp(self.xmasssave), _vp(self.rho_rel),
                                            C.byref(nrel)), "fpx_releaseparticles")
        self.n, self.numparticlecount = int(n.value), int(npc.value)
        return int(nrel.value)

    def split_particles(self, itime):
        n = C.c_int64(self.n)
        check(self.lib.fpx_split_particles(self.h, int(itime), C.byref(n)), "fpx_split_particles")
        self.n = int(n.value)

    def set_numpart(self, n):
        """numpart (com_mod.f90:676) as the host sees it; a smaller value than the engine's puts a locality-sorted cloud back
        into particle-number order first (the spaces beyond numpart must be vacant)."""
        check(self.lib.fpx_set_numpart(self.h, int(n)), "fpx_set_numpart")
        self.n = int(n)

    # ---- mpi_mod.f90:566-856: levelling the ranks' particle counts (the host keeps the transport) ----
    def redist_pack(self, itime, num_trans):
        """Sender: the last num_trans storage spaces -> one message (numpy uint8 array); they are terminated, self.n shrinks."""
        nb = int(self.lib.fpx_redist_bytes(self.h, int(num_trans)))
        buf = np.zeros(max(nb, 1), np.uint8)
        n = C.c_int64(self.n)
        check(self.lib.fpx_redist_pack(self.h, int(itime), int(num_trans), _vp(buf), nb, C.byref(n)), "fpx_redist_pack")
        self.n = int(n.value)
        return buf[:nb]

    def redist_unpack(self, itime, num_trans, buf):
        """Receiver: places the message's particles that are alive at itime into vacant storage spaces; self.n follows."""
        buf = np.ascontiguousarray(buf, np.uint8)
        n = C.c_int64(self.n)
        check(self.lib.fpx_redist_unpack(self.h, int(itime), int(num_trans), _vp(buf), buf.size, C.byref(n)), "fpx_redist_unpack")
        self.n = int(n.value)

    def set_windtime(self, memtime, memind):
        mt = (C.c_int32 * 2)(int(memtime[0]), int(memtime[1]))
        mi = (C.c_int32 * 2)(int(memind[0]), int(memind[1]))
        check(self.lib.fpx_set_windtime(self.h, mt, mi), "fpx_set_windtime")

    # ---- particles ----------------------------------------------------------
    def upload_particles_from_scenario(self, sc, first=0):
        n = int(sc["npart"])
        rt = self.hreal
        keep = {}
        p = FpxParticles()

        def put(name, key, dtype):
            if key in sc:
                keep[name] = np.ascontiguousarray(np.asarray(sc[key]).astype(dtype))
                setattr(p, name, keep[name].ctypes.data)
        put("xtra1", "xtra1", np.float64); put("ytra1", "ytra1", np.float64); put("ztra1", "ztra1", rt)
        for k in ("uap", "ucp", "uzp", "us", "vs", "ws"):
            put(k, k, rt)
        for k in ("itra1", "itramem", "idt", "npoint", "nclass", "itrasplit"):
            put(k, k, np.int32)
        put("cbt", "cbt", np.int16)
        if "xmass1" in sc:
            keep["xmass1"] = np.ascontiguousarray(np.asarray(sc["xmass1"]).astype(rt).reshape(self.nspec, n))
            p.xmass1 = keep["xmass1"].ctypes.data
            p.xmass1_ld = n
        if "xscav_frac1" in sc:
            keep["xscav_frac1"] = np.ascontiguousarray(np.asarray(sc["xscav_frac1"]).astype(rt).reshape(self.nspec, n))
            p.xscav_frac1 = keep["xscav_frac1"].ctypes.data
            p.xmass1_ld = n
        check(self.lib.fpx_upload_particles(self.h, first, n, C.byref(p)), "fpx_upload_particles")
        self.n = max(self.n, first + n)

    def seed_particles(self, n, seed=0x5EED, frac_pbl=0.5, zmax=12000.0, lat_margin_cells=None, itime0=0):
        if lat_margin_cells is None:
            lat_margin_cells = 0.03 * (self.ny - 1)
        check(self.lib.fpx_seed_particles(self.h, n, seed, frac_pbl, zmax, lat_margin_cells, itime0),
              "fpx_seed_particles")
        self.n = n

    def download(self, first=0, count=None):
        n = self.n - first if count is None else count
        rt = self.hreal
        out = dict(xtra1=np.empty(n, np.float64), ytra1=np.empty(n, np.float64), ztra1=np.empty(n, rt),
                   uap=np.empty(n, rt), ucp=np.empty(n, rt), uzp=np.empty(n, rt), us=np.empty(n, rt),
                   vs=np.empty(n, rt), ws=np.empty(n, rt), itra1=np.empty(n, np.int32),
                   itramem=np.empty(n, np.int32), idt=np.empty(n, np.int32), npoint=np.empty(n, np.int32),
                   nclass=np.empty(n, np.int32), itrasplit=np.empty(n, np.int32), cbt=np.empty(n, np.int16),
                   xmass1=np.empty((self.nspec, n), rt))
        if self.bkdep:
            out["xscav_frac1"] = np.empty((self.nspec, n), rt)
        p = FpxParticles()
        for k, a in out.items():
            setattr(p, k, a.ctypes.data)
        p.xmass1_ld = n
        check(self.lib.fpx_download_particles(self.h, first, n, C.byref(p)), "fpx_download_particles")
        res = {k: (v.astype(np.float64) if v.dtype.kind == "f" else v.astype(np.int32)) for k, v in out.items()}
        return res

    # ---- output grids ---------------------------------------------------------
    def outgrid_from_scenario(self, sc):
        """OUTGRID state as readoutgrid.f90 / outgrid_init.f90 leave it."""
        rt = self.hreal
        nxg, nyg, nzg = (int(v) for v in sc["outgrid"])
        dxo, dyo, lon0, lat0 = (float(v) for v in sc["outgeom"])
        g = FpxOutgrid()
        g.struct_bytes = C.sizeof(FpxOutgrid)
        g.numxgrid, g.numygrid, g.numzgrid = nxg, nyg, nzg
        g.dxout, g.dyout = float(rt(dxo)), float(rt(dyo))
        # xoutshift=xlon0-outlon0 in the host's real kind (readoutgrid.f90:199-200)
        g.xoutshift = float(rt(self.cfg.xlon0) - rt(lon0))
        g.youtshift = float(rt(self.cfg.ylat0) - rt(lat0))
        lage = np.asarray(sc["lage"]).ravel()
        iofr = int(np.asarray(sc["concflags"]).ravel()[1])
        mps = int(sc.get("numpoint", 1)) if iofr == 1 else 1      # maxpointspec_act, readreleases.f90 / outgrid_init.f90
        ncu = int(sc.get("nclassunc", 1))
        g.maxpointspec_act, g.nclassunc = mps, ncu
        g.nageclass = len(lage)
        for i, v in enumerate(lage):
            g.lage[i] = int(v)
        g.ind_samp, g.ioutputforeachrelease = (int(v) for v in sc["concflags"])
        g.lusekerneloutput = int(sc.get("lusekerneloutput", 1))   # par_mod.f90:39 (a compile-time switch of the host)
        oh = np.ascontiguousarray(np.asarray(sc["outheight"]).astype(rt))
        check(self.lib.fpx_outgrid_init(self.h, C.byref(g), _vp(oh)), "fpx_outgrid_init")
        if "outtimes" in sc:
            check(self.lib.fpx_set_output_times(self.h, int(sc["outtimes"][0]), int(sc["outtimes"][1])),
                  "fpx_set_output_times")
        self.gshape = (len(lage), ncu, mps, self.nspec, nzg, nyg, nxg)
        if "outgridn" in sc:          # OUTGRID_NEST (readoutgrid_nest.f90)
            nxn, nyn = (int(v) for v in sc["outgridn"])
            dxn, dyn, lon0n, lat0n = (float(v) for v in sc["outgeomn"])
            gn = FpxOutgridNest()
            gn.struct_bytes = C.sizeof(FpxOutgridNest)
            gn.numxgridn, gn.numygridn = nxn, nyn
            gn.dxoutn, gn.dyoutn = float(rt(dxn)), float(rt(dyn))
            gn.xoutshiftn = float(rt(self.cfg.xlon0) - rt(lon0n))
            gn.youtshiftn = float(rt(self.cfg.ylat0) - rt(lat0n))
            check(self.lib.fpx_outgrid_nest_init(self.h, C.byref(gn)), "fpx_outgrid_nest_init")
            self.gshape_nest = (len(lage), ncu, mps, self.nspec, nzg, nyn, nxn)
        self.nreceptor = 0
        if "receptors" in sc:         # RECEPTORS (readreceptors.f90): x, y in grid coordinates, cell area
            r = np.asarray(sc["receptors"], dtype=np.float64).reshape(3, -1)
            self.nreceptor = r.shape[1]
            rx, ry, ra = (np.ascontiguousarray(r[k].astype(rt)) for k in range(3))
            check(self.lib.fpx_receptors_init(self.h, self.nreceptor, _vp(rx), _vp(ry), _vp(ra)), "fpx_receptors_init")

    def wet_from_scenario(self, sc):
        """Wet-scavenging species parameters (readspecies.f90) + precipitation/cloud fields."""
        w = FpxWetConfig()
        w.struct_bytes = C.sizeof(FpxWetConfig)
        for i in range(self.nspec):
            w.wetdepspec[i] = int(np.asarray(sc["wetdepspec"]).ravel()[i])
            for k in ("weta_gas", "wetb_gas", "crain_aero", "csnow_aero", "ccn_aero", "in_aero", "henry"):
                getattr(w, k)[i] = float(np.asarray(sc[k]).ravel()[i])
        w.readclouds = 0
        check(self.lib.fpx_wet_init(self.h, C.byref(w)), "fpx_wet_init")
        for m in (0, 1):
            keep = {}
            f = FpxWetFields()
            for k in ("lsprec", "convprec", "tcc"):
                keep[k] = self._host2(sc[k], m)
                setattr(f, k, keep[k].ctypes.data)
            keep["tt"] = self._host3(sc["tt"], m)
            f.tt = keep["tt"].ctypes.data
            c8 = np.zeros((self.nzmax, self.nymax, self.nxmax), np.int8)
            c8[: self.nz, : self.ny, : self.nx] = np.asarray(sc["clouds"])[m]
            keep["clouds"] = c8
            f.clouds = c8.ctypes.data
            ch = np.zeros((self.nymax, self.nxmax), np.int32)
            ch[: self.ny, : self.nx] = np.asarray(sc["cloudsh"])[m]
            keep["cloudsh"] = ch
            f.cloudsh = ch.ctypes.data
            check(self.lib.fpx_upload_wet_fields(self.h, m + 1, C.byref(f)), "fpx_upload_wet_fields")
        if "lsprecn" in sc:      # the nest's own fields (compact: nxmaxn = nxn, nymaxn = nyn as in upload_nests_from_scenario)
            rt = self.hreal
            for m in (0, 1):
                keep = {}
                f = FpxWetFields()
                for k, kn in (("lsprec", "lsprecn"), ("convprec", "convprecn"), ("tcc", "tccn"), ("tt", "ttn")):
                    a = np.asarray(sc[kn])[m]
                    if kn == "ttn" and self.nzmax != self.nz:
                        b = np.zeros((self.nzmax,) + a.shape[1:]); b[: self.nz] = a; a = b
                    keep[k] = np.ascontiguousarray(a.astype(rt))
                    setattr(f, k, keep[k].ctypes.data)
                cn = np.asarray(sc["cloudsn"])[m]
                c8 = np.zeros((self.nzmax,) + cn.shape[1:], np.int8)
                c8[: self.nz] = cn
                keep["clouds"] = c8
                f.clouds = c8.ctypes.data
                check(self.lib.fpx_upload_wet_nest_fields(self.h, 1, m + 1, C.byref(f), 0), "fpx_upload_wet_nest_fields")

    def wetdepo(self, itime=None, ltsample=None, loutnext=None):
        itime = self.itime if itime is None else itime
        ltsample = self.lsynctime if ltsample is None else ltsample
        if loutnext is None:
            loutnext = int(self.sc["outtimes"][0]) if "outtimes" in self.sc else 0
        check(self.lib.fpx_wetdepo(self.h, int(itime), int(ltsample), int(loutnext)), "fpx_wetdepo")

    def wetgrid(self, allreduce=False):
        na, nc, mp, nsp, nzg, nyg, nxg = self.gshape
        d = np.empty((na, nc, mp, nsp, nyg, nxg), np.float32)
        check(self.lib.fpx_get_wetgrid(self.h, _vp(d), int(allreduce)), "fpx_get_wetgrid")
        return d.astype(np.float64)

    def conccalc(self, itime=None, weight=1.0):
        check(self.lib.fpx_conccalc(self.h, int(self.itime if itime is None else itime), float(weight)),
              "fpx_conccalc")

    def grids(self, allreduce=False, clear=False):
        """-> (gridunc, drygridunc) as float64 arrays shaped (age, class, pointspec, spec, z, y, x) /
        (age, class, pointspec, spec, y, x)."""
        na, nc, mp, nsp, nzg, nyg, nxg = self.gshape
        g = np.empty((na, nc, mp, nsp, nzg, nyg, nxg), self.hreal)
        d = np.empty((na, nc, mp, nsp, nyg, nxg), np.float32)
        check(self.lib.fpx_get_grids(self.h, _vp(g), _vp(d), int(allreduce), int(clear)), "fpx_get_grids")
        return g.astype(np.float64), d.astype(np.float64)

    def grids_nest(self, allreduce=False, clear=False):
        """-> (griduncn, drygriduncn, wetgriduncn) of the nested output grid, float64."""
        na, nc, mp, nsp, nzg, nyg, nxg = self.gshape_nest
        g = np.empty((na, nc, mp, nsp, nzg, nyg, nxg), self.hreal)
        d = np.empty((na, nc, mp, nsp, nyg, nxg), np.float32)
        w = np.empty((na, nc, mp, nsp, nyg, nxg), np.float32)
        check(self.lib.fpx_get_grids_nest(self.h, _vp(g), _vp(d), _vp(w), int(allreduce), int(clear)), "fpx_get_grids_nest")
        return g.astype(np.float64), d.astype(np.float64), w.astype(np.float64)

    def receptors(self, allreduce=False, clear=False, ld=20):
        """-> creceptor as float64 (spec, receptor); the host array is creceptor(ld=maxreceptor, maxspec)."""
        ld = max(ld, self.nreceptor)
        c = np.zeros((self.nspec, ld), self.hreal)
        check(self.lib.fpx_get_receptors(self.h, _vp(c), ld, int(allreduce), int(clear)), "fpx_get_receptors")
        return c[:, :self.nreceptor].astype(np.float64)

    def comm_unique_id(self):
        buf = (C.c_char * 128)()
        check(self.lib.fpx_comm_unique_id(buf, 128), "fpx_comm_unique_id")
        return bytes(buf)

    def comm_init_host(self, dist, nranks, rank):
        """fpx_comm_init_host with a torch.distributed process group as the transport (what an MPI host does with
        MPI_Allreduce): the engine hands host buffers to the callback, which sums them over the ranks."""
        import torch

        def _allreduce(user, send, recv, count, dtype):
            try:
                ct = C.c_double if dtype == 1 else C.c_float
                a = np.ctypeslib.as_array(C.cast(send, C.POINTER(ct)), shape=(count,))
                t = torch.from_numpy(np.array(a, copy=True))
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                np.ctypeslib.as_array(C.cast(recv, C.POINTER(ct)), shape=(count,))[:] = t.numpy()
                return 0
            except Exception:      # never let an exception cross the C boundary
                return 1
        self._allreduce_cb = _lib.ALLREDUCE_FN(_allreduce)      # keep the thunk alive as long as the engine
        check(self.lib.fpx_comm_init_host(self.h, int(nranks), int(rank), self._allreduce_cb, None), "fpx_comm_init_host")

    def count_particles(self, allreduce=False):
        """(live, numpart) of this rank and summed over the ranks: the particle count the reference's root reduces at
        every output time (timemanager_mpi.f90:552-562)."""
        loc = (C.c_int64 * 2)()
        tot = (C.c_int64 * 2)()
        check(self.lib.fpx_count_particles(self.h, loc, tot, int(allreduce)), "fpx_count_particles")
        return (int(loc[0]), int(loc[1])), (int(tot[0]), int(tot[1]))

    def set_option(self, name, value):
        """fpx_set_option: a tuning / diagnostic knob of this handle (include/flexpart_amd.h lists the names)."""
        check(self.lib.fpx_set_option(self.h, str(name).encode(), str(value).encode()), f"fpx_set_option({name})")

    def info(self, name):
        """fpx_get_info: what the engine decided or did ("time_blended_packs", "blended_steps", "pbl_launches_per_step", ...)."""
        v = C.c_int64(0)
        check(self.lib.fpx_get_info(self.h, str(name).encode(), C.byref(v)), f"fpx_get_info({name})")
        return int(v.value)

    def lane_stats(self, reset=False):
        """Per code region of the Langevin kernel: (executions by a wave, mean active lanes); zeros unless the library was
        built with -DFPX_LANE_STATS."""
        out = (C.c_uint64 * 32)()
        check(self.lib.fpx_lane_stats(self.h, out, 32, int(reset)), "fpx_lane_stats")
        names = ("pass", "substep", "cbl", "gauss_cblflag", "exp_form", "hs_neutral", "hs_unstable", "hs_stable", "refill", "handover", "outer_loop")
        d = {nm: (int(out[2 * i]), (out[2 * i + 1] / out[2 * i]) if out[2 * i] else 0.0) for i, nm in enumerate(names)}
        nw = int(out[22])
        if nw:   # timeline of the persistent waves, all launches together (100 MHz ticks -> ms)
            ms = 1e-5
            d["timeline_ms"] = (nw, {"mean_start_to_list_exhausted": out[23] / nw * ms, "mean_start_to_end": out[24] / nw * ms,
                                     "longest_wave": out[25] * ms, "first_exhaustion": ((~out[26]) & (2**64 - 1)) * ms,
                                     "longest_drain": out[27] * ms})
        return d

    def comm_init(self, uid, nranks, rank):
        buf = (C.c_char * 128).from_buffer_copy(uid)
        check(self.lib.fpx_comm_init(self.h, buf, 128, int(nranks), int(rank)), "fpx_comm_init")

    # ---- the loop body --------------------------------------------------------
    def step(self, itime=None):
        """One pass of the particle loop at `itime` (default: the engine's clock)."""
        if itime is None:
            itime = self.itime
        st = FpxStepStats()
        check(self.lib.fpx_step(self.h, int(itime), C.byref(st)), "fpx_step")
        self.itime = int(itime) + self.lsynctime
        return {k: getattr(st, k) for k, _ in FpxStepStats._fields_}

    def step_async(self, itime=None):
        if itime is None:
            itime = self.itime
        check(self.lib.fpx_step_async(self.h, int(itime)), "fpx_step_async")
        self.itime = int(itime) + self.lsynctime

    def sync(self):
        check(self.lib.fpx_sync(self.h), "fpx_sync")

    def counters(self, reset=False):
        st = FpxStepStats()
        check(self.lib.fpx_counters(self.h, C.byref(st), int(reset)), "fpx_counters")
        return {k: getattr(st, k) for k, _ in FpxStepStats._fields_}

    def kernel_time(self, reset=False):
        ms = C.c_double(0)
        ln = C.c_int64(0)
        check(self.lib.fpx_kernel_time(self.h, C.byref(ms), C.byref(ln), int(reset)), "fpx_kernel_time")
        return ms.value, ln.value

    def kernel_times(self, reset=False):
        """(k_prep + work-list sort, k_pbl_loop, k_pbl_finish, k_prep alone) cumulative device ms and the number of steps."""
        ms = (C.c_double * 4)()
        ln = C.c_int64(0)
        check(self.lib.fpx_kernel_times(self.h, ms, C.byref(ln), int(reset)), "fpx_kernel_times")
        return list(ms), ln.value

    def sort(self):
        check(self.lib.fpx_sort_particles(self.h), "fpx_sort_particles")

    def rannumb(self):
        a = np.empty(1000000, self.hreal)
        check(self.lib.fpx_rng_get_table(self.h, _vp(a), 1000000), "fpx_rng_get_table")
        return a

    def run(self, nsteps=None):
        out = []
        for _ in range(int(self.sc["nsteps"]) if nsteps is None else nsteps):
            if self.has_wet and self.itime != 0:     # wetdepo first, timemanager.f90:164-169
                self.wetdepo()
            if self.has_conv:                        # timemanager.f90:258-262 (lconvection = 1)
                self.convmix(self.itime)
            self.step()
            if self.gshape is not None:
                self.conccalc(self.itime, 1.0)     # sample at the new positions (conccalc.f90)
            out.append(self.download())
        return out

    def close(self):
        if self.h:
            self.lib.fpx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


__all__ = ["Engine", "RNG_TABLE_SEQ", "RNG_TABLE_COUNTER", "RNG_PHILOX"]
