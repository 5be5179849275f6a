"""TEST INFRASTRUCTURE ONLY -- scenario/record files for oracle/_ref/flexref_r{4,8}.

Record stream: {name char[16], dtype int32 (1=int32, 2=float64), count int64,
payload}, closed by an 'END' record.  Written for / read back from our own
Fortran driver oracle/ref_driver.f90.
"""
from __future__ import annotations

import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# order matters: 'grid' and 'nspec' before fields, 'npart' before particle arrays
_ORDER = ["grid", "geom", "globalflags", "height", "nmixz", "memtime", "memind", "ldirect",
          "lsynctime", "method", "mintime", "ctl", "ifine", "turbswitch", "cblflag",
          "mdomainfill", "lsettling", "nspec", "drydep", "drydepspec", "density", "dquer",
          "vsetaver", "cunningham", "decay", "turbpar", "lage", "nclassunc", "mquasilag", "nsteps", "itime0",
          "outgrid", "outgeom", "outheight", "outgridn", "outgeomn", "receptors", "concflags", "outtimes",
          "wetdep", "wetdepspec", "weta_gas", "wetb_gas", "crain_aero", "csnow_aero", "ccn_aero", "in_aero", "henry",
          "uu", "vv", "ww", "rho", "drhodz", "tt", "uupol", "vvpol",
          "hmix", "ustar", "wstar", "oli", "tropopause", "vdep",
          "nest", "nestgeom", "uun", "vvn", "wwn", "rhon", "drhodzn", "hmixn", "ustarn", "wstarn", "olin",
          "tropopausen", "vdepn", "lsprecn", "convprecn", "tccn", "ttn", "cloudsn", "cloudshn", "lsprec", "convprec", "tcc", "clouds", "cloudsh",
          "numpoint", "npart", "xtra1", "ytra1", "ztra1", "itra1", "itramem", "npoint", "nclass", "idt",
          "uap", "ucp", "uzp", "us", "vs", "ws", "cbt", "xmass1", "xmass", "npart_rel",
          "drybkdep", "wetbkdep", "zpoint1", "zpoint2"]      # backward runs with receptor scavenging: after 'numpoint' and 'npart' 
_INT = {"grid", "globalflags", "nmixz", "memtime", "memind", "ldirect", "lsynctime", "method",
        "mintime", "ifine", "turbswitch", "cblflag", "mdomainfill", "lsettling", "nspec",
        "drydep", "drydepspec", "lage", "nclassunc", "mquasilag", "numpoint", "npart_rel", "nsteps", "itime0", "npart", "itra1", "itramem",
        "npoint", "nclass", "idt", "cbt", "outgrid", "outgridn", "concflags", "outtimes", "wetdep", "wetdepspec", "clouds", "cloudsh", "cloudsn", "cloudshn", "nest",
        "drybkdep", "wetbkdep"}


# scenario key -> (value as shipped, suffix of the flang build with the other value: oracle/build_ref.sh)
_COMPILED = {"turboff": (0, "t"), "interpolhmix": (0, "h"), "lusekerneloutput": (1, "k")}


def write_scenario(path, sc):
    with open(path, "wb") as fh:
        for name in _ORDER:
            if name not in sc:
                continue
            v = sc[name]
            if name in _INT:
                a = np.ascontiguousarray(np.asarray(v, dtype=np.int32).ravel())
                code = 1
            else:
                a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel())
                code = 2
            fh.write(struct.pack("<16siq", name.encode().ljust(16), code, a.size))
            fh.write(a.tobytes())
        fh.write(struct.pack("<16siq", b"END".ljust(16), 1, 0))
    # (turboff, interpolhmix, lusekerneloutput are compile-time parameters of the reference: not in the file, but the
    # binary must be the variant built with them -- run_reference checks)
    unknown = set(sc) - set(_ORDER) - {"par_nxmax", "particle_base"} - set(_COMPILED)
    if unknown:
        raise KeyError(f"scenario keys not understood by the reference driver: {sorted(unknown)}")


def read_records(path):
    """-> list of (name, ndarray) in file order."""
    out = []
    with open(path, "rb") as fh:
        while True:
            hdr = fh.read(28)
            if len(hdr) < 28:
                break
            name, code, cnt = struct.unpack("<16siq", hdr)
            name = name.decode().strip()
            if name == "END":
                break
            dt = np.int32 if code == 1 else np.float64
            a = np.frombuffer(fh.read(cnt * np.dtype(dt).itemsize), dtype=dt).copy()
            out.append((name, a))
    return out


def ref_binary(kind):
    return os.path.join(HERE, "_ref", f"flexref_{kind}")


def have_ref(kind="r8"):
    return os.access(ref_binary(kind), os.X_OK)


def run_reference(sc, kind="r8", workdir="/tmp", timing=False, tag="scen", gpu=False):
    """Run the compiled reference on a scenario -> dict(steps=[{...}], rannumb=..., ...)."""
    os.makedirs(workdir, exist_ok=True)
    for key, (shipped, suffix) in _COMPILED.items():
        if (int(sc.get(key, shipped)) != shipped) != kind.endswith(suffix):
            raise ValueError(f"scenario {key} = {sc.get(key, shipped)} needs the reference build "
                             f"{'with' if int(sc.get(key, shipped)) != shipped else 'without'} suffix '{suffix}', not {kind}")
    fs = os.path.join(workdir, f"{tag}_{os.getpid()}.scen")
    fo = os.path.join(workdir, f"{tag}_{os.getpid()}.out")
    write_scenario(fs, sc)
    # gpu = True: the Fortran host drives the engine (fp64 arithmetic); gpu = 32: the reference-typed f32 engine
    cmd = f"ulimit -s unlimited; exec {ref_binary(kind)} {fs} {fo}" + (" timing" if timing else "") + (" gpu32" if gpu == 32 else (" gpu" if gpu else ""))
    res = subprocess.run(["bash", "-c", cmd], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"reference driver failed: {res.stdout}\n{res.stderr}")
    recs = read_records(fo)
    os.remove(fs)
    os.remove(fo)
    nspec = int(np.asarray(sc.get("nspec", 1)).ravel()[0])
    out = {"steps": [], "stdout": res.stdout}
    cur = None
    for name, a in recs:
        if name in ("rannumb", "northpolemap", "southpolemap", "derived", "timing", "gridunc", "drygridunc", "wetgridunc", "griduncn", "drygriduncn", "wetgriduncn", "creceptor"):
            out[name] = a
            continue
        if name == "xtra1":
            cur = {}
            out["steps"].append(cur)
        if name in ("xmass1", "xscav_frac1"):
            cur.setdefault(name, []).append(a)
            if len(cur[name]) == nspec:
                cur[name] = np.stack(cur[name])
        else:
            cur[name] = a
    return out


# --------------------------------------------------------------------------
# verttransform_ecmwf through oracle/_ref/vtref_rK (oracle/ref_vt_driver.f90)
# --------------------------------------------------------------------------
_VT_ORDER = ["grid", "geom", "globalflags", "ncalls", "akz", "bkz", "aknew", "bknew", "ps", "tt2", "td2",
             "tth", "qvh", "uuh", "vvh", "pvh", "wwh"]


def have_vt_ref(kind="r8"):
    return os.access(os.path.join(HERE, "_ref", f"vtref_{kind}"), os.X_OK)


def run_vt_reference(m, kind="r8", workdir="/tmp", ncalls=1, gpu=False, nest=None):
    """The unmodified verttransform_ecmwf on a synthetic.model_levels() dict -> dict of fields [nz][ny][nx]."""
    os.makedirs(workdir, exist_ok=True)
    fs = os.path.join(workdir, f"vt_{os.getpid()}.scen")
    fo = os.path.join(workdir, f"vt_{os.getpid()}.out")
    mm = dict(m, ncalls=ncalls)
    with open(fs, "wb") as fh:
        for name in _VT_ORDER:
            v = mm[name]
            if name in ("grid", "globalflags", "ncalls"):
                a = np.ascontiguousarray(np.asarray(v, dtype=np.int32).ravel()); code = 1
            else:
                a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel()); code = 2
            fh.write(struct.pack("<16siq", name.encode().ljust(16), code, a.size))
            fh.write(a.tobytes())
        if nest is not None:      # one nest (kind r8n: the reference's par_mod_meteoswiss.f90 has maxnests=1)
            recs = [("nestgrid", nest["grid"][:2], 1), ("nestgeom", nest["geom"], 2)]
            recs += [(k + "n", nest[k], 2) for k in ("ps", "tt2", "td2", "tth", "qvh", "uuh", "vvh", "pvh", "wwh")]
            for name, v, code in recs:
                a = np.ascontiguousarray(np.asarray(v, dtype=np.int32 if code == 1 else np.float64).ravel())
                fh.write(struct.pack("<16siq", name.encode().ljust(16), code, a.size))
                fh.write(a.tobytes())
        fh.write(struct.pack("<16siq", b"END".ljust(16), 1, 0))
    exe = os.path.join(HERE, "_ref", f"vtref_{kind}")
    res = subprocess.run(["bash", "-c", f"ulimit -s unlimited; exec {exe} {fs} {fo}" + (" gpu" if gpu else "")], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"reference verttransform driver failed: {res.stdout}\n{res.stderr}")
    nx, ny, nz = (int(v) for v in m["grid"])
    out = {}
    nn = (int(nest["grid"][0]), int(nest["grid"][1])) if nest is not None else (0, 0)
    for name, a in read_records(fo):
        if nest is not None and name in ("uun", "vvn", "wwn", "ttn", "qvn", "pvn", "rhon", "drhodzn"):
            out[name] = a.reshape(nz, nn[1], nn[0])
        else:
            out[name] = a.reshape(nz, ny, nx) if a.size == nx * ny * nz else a
    os.remove(fs)
    os.remove(fo)
    out["nmixz"] = int(out["nmixz"][0])
    return out


# --------------------------------------------------------------------------
# partoutput through oracle/_ref/poref_rK (oracle/ref_po_driver.f90)
# --------------------------------------------------------------------------
_PO_ORDER = ["grid", "geom", "height", "memtime", "memind", "nspec", "itime", "oro", "pv", "qv", "tt", "rho",
             "hmix", "tropopause", "npart", "xtra1", "ytra1", "ztra1", "itra1", "itramem", "npoint", "xmass1"]
_PO_INT = {"grid", "memtime", "memind", "nspec", "itime", "npart", "itra1", "itramem", "npoint"}


def have_po_ref(kind="r8"):
    return os.access(os.path.join(HERE, "_ref", f"poref_{kind}"), os.X_OK)


def run_po_reference(sc, kind="r8", workdir="/tmp", gpu=False):
    """The unmodified partoutput on a scenario dict -> bytes of the file partposit_end it wrote."""
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="po_", dir=workdir)
    try:
        fs = os.path.join(d, "po.scen")
        with open(fs, "wb") as fh:
            for name in _PO_ORDER:
                v = sc[name]
                if name in _PO_INT:
                    a = np.ascontiguousarray(np.asarray(v, dtype=np.int32).ravel()); code = 1
                else:
                    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel()); code = 2
                fh.write(struct.pack("<16siq", name.encode().ljust(16), code, a.size))
                fh.write(a.tobytes())
            fh.write(struct.pack("<16siq", b"END".ljust(16), 1, 0))
        exe = os.path.join(HERE, "_ref", f"poref_{kind}")
        res = subprocess.run(["bash", "-c", f"ulimit -s unlimited; exec {exe} {fs} {d}/" + (" gpu" if gpu else "")], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"reference partoutput driver failed: {res.stdout}\n{res.stderr}")
        with open(os.path.join(d, "partposit_end"), "rb") as fh:
            return fh.read()
    finally:
        shutil.rmtree(d, ignore_errors=True)


# --------------------------------------------------------------------------
# readpartpositions through oracle/_ref/rpref_rK (oracle/ref_rp_driver.f90)
# --------------------------------------------------------------------------
def have_rp_ref(kind="r8"):
    return os.access(os.path.join(HERE, "_ref", f"rpref_{kind}"), os.X_OK)


def run_rp_reference(file_bytes, rs, kind="r8", workdir="/tmp", gpu=False):
    """The unmodified readpartpositions on a dump (bytes of partposit_end) -> dict of the arrays it filled."""
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="rp_", dir=workdir)
    try:
        with open(os.path.join(d, "partposit_end"), "wb") as fh:
            fh.write(file_bytes)
        fs = os.path.join(d, "rp.scen")
        with open(fs, "wb") as fh:
            for name, code, dt in (("geom", 2, np.float64), ("nspec", 1, np.int32), ("restart", 1, np.int32)):
                a = np.ascontiguousarray(np.asarray(rs[name], dtype=dt).ravel())
                fh.write(struct.pack("<16siq", name.encode().ljust(16), code, a.size))
                fh.write(a.tobytes())
            fh.write(struct.pack("<16siq", b"END".ljust(16), 1, 0))
        exe = os.path.join(HERE, "_ref", f"rpref_{kind}")
        fo = os.path.join(d, "out.bin")
        res = subprocess.run(["bash", "-c", f"ulimit -s unlimited; exec {exe} {fs} {d}/ {fo}" + (" gpu" if gpu else "")], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"reference readpartpositions driver failed: {res.stdout}\n{res.stderr}")
        out = {}
        xm = []
        for name, a in read_records(fo):
            if name == "xmass1":
                xm.append(a)
            elif name == "numpart":
                out["numpart"], out["numparticlecount"] = int(a[0]), int(a[1])
            else:
                out[name] = a
        out["xmass1"] = np.stack(xm)
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


# --------------------------------------------------------------------------
# concoutput through oracle/_ref/coref_r4 (oracle/ref_co_driver.f90)
# --------------------------------------------------------------------------
def have_co_ref(kind="r4"):
    return os.access(os.path.join(HERE, "_ref", f"coref_{kind}"), os.X_OK)


def run_co_reference(co, kind="r4", workdir="/tmp", nest=False):
    """The unmodified concoutput on a dict (outgrid, outgeom, outheight, area, volume, gridunc[, wetgridunc, drygridunc])
    -> {file name: bytes} of the grid_conc_* files it wrote."""
    import glob
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="co_", dir=workdir)
    try:
        fs = os.path.join(d, "co.scen")
        with open(fs, "wb") as fh:
            for name in ("outgrid", "outgeom", "outheight", "iout", "met", "metgeom", "height", "weightmolar", "rho2",
                         "area", "volume", "classes", "gridunc", "wetgridunc", "drygridunc"):
                if name not in co:
                    continue
                code = 1 if name in ("outgrid", "iout", "met", "classes") else 2
                a = np.ascontiguousarray(np.asarray(co[name], dtype=np.int32 if code == 1 else np.float64).ravel())
                fh.write(struct.pack("<16siq", name.encode().ljust(16), code, a.size))
                fh.write(a.tobytes())
            fh.write(struct.pack("<16siq", b"END".ljust(16), 1, 0))
        exe = os.path.join(HERE, "_ref", f"coref_{kind}")
        res = subprocess.run(["bash", "-c", f"ulimit -s unlimited; exec {exe} {fs} {d}/" + (" nest" if nest else "")], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"reference concoutput driver failed: {res.stdout}\n{res.stderr}")
        return {os.path.basename(f): open(f, "rb").read()
                for f in sorted(glob.glob(os.path.join(d, "grid_conc_*")) + glob.glob(os.path.join(d, "grid_pptv_*")))}
    finally:
        shutil.rmtree(d, ignore_errors=True)


# --------------------------------------------------------------------------
# releaseparticles + splitting through oracle/_ref/relref_rK (oracle/ref_rel_driver.f90)
# --------------------------------------------------------------------------
_REL_ORDER = ["grid", "geom", "xglobal", "height", "nspec", "bdate", "switches", "times", "oro", "rho2", "tt2",
              "par_nxmax", "nest", "nestcorners", "oron", "rhon2", "ttn2",
              "numpoint", "ireleasestart", "ireleaseend", "npart_rel", "kindz", "xpoint1", "xpoint2", "ypoint1", "ypoint2",
              "zpoint1", "zpoint2", "xmass", "point_hour", "area_hour", "point_dow", "area_dow",
              "npart", "xtra1", "ytra1", "ztra1", "itra1", "itramem", "itrasplit", "npoint", "nclass", "idt", "uap", "xmass1"]
_REL_INT = {"par_nxmax", "nest", "grid", "xglobal", "nspec", "bdate", "switches", "times", "numpoint", "ireleasestart", "ireleaseend", "npart_rel", "kindz",
            "npart", "itra1", "itramem", "itrasplit", "npoint", "nclass", "idt"}


def have_rel_ref(kind="r8", nest=False):
    return os.access(os.path.join(HERE, "_ref", f"relref_{kind}" + ("n" if nest else "")), os.X_OK)


def run_rel_reference(rs, kind="r8", workdir="/tmp", gpu=False):
    """The unmodified releaseparticles (+ our restatement of the splitting block) on a synthetic.release_case() dict
    -> list of per-call dicts (state = [itime, numpart, numparticlecount], the particle arrays 1..numpart,
    xmasssave, rho_rel)."""
    import tempfile
    os.makedirs(workdir, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="rel_", dir=workdir) as d:
        fs, fo = os.path.join(d, "rel.scen"), os.path.join(d, "rel.out")
        with open(fs, "wb") as fh:
            for name in _REL_ORDER:
                if name not in rs:
                    continue
                code = 1 if name in _REL_INT else 2
                a = np.ascontiguousarray(np.asarray(rs[name], dtype=np.int32 if code == 1 else np.float64).ravel())
                fh.write(struct.pack("<16siq", name.encode().ljust(16), code, a.size))
                fh.write(a.tobytes())
            fh.write(struct.pack("<16siq", b"END".ljust(16), 1, 0))
        exe = os.path.join(HERE, "_ref", f"relref_{kind}" + ("n" if "nest" in rs else ""))
        res = subprocess.run(["bash", "-c", f"ulimit -s unlimited; exec {exe} {fs} {fo}" + (" gpu" if gpu else "")], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"reference releaseparticles driver failed: {res.stdout}\n{res.stderr}")
        nspec = int(rs["nspec"])
        calls, cur = [], None
        for name, a in read_records(fo):
            if name == "state":
                cur = {"state": a, "xmass1": []}
                calls.append(cur)
            elif name == "xmass1":
                cur["xmass1"].append(a)
            else:
                cur[name] = a
        for c in calls:
            c["xmass1"] = np.stack(c["xmass1"]) if c["xmass1"] else np.zeros((nspec, 0))
        return calls


# --------------------------------------------------------------------------
# the leaf routines of calcpar through oracle/_ref/cpref_rK (oracle/ref_cp_driver.f90)
# --------------------------------------------------------------------------
def have_cp_ref(kind="r8"):
    return os.access(os.path.join(HERE, "_ref", f"cpref_{kind}"), os.X_OK)


def run_cp_leaf_reference(ps, t, td, stress, kind="r8", workdir="/tmp"):
    """The unmodified scalev, ew, f_qvsat on arrays of arguments -> (n,3) float64: scalev(ps,t,td,stress), ew(td), f_qvsat(ps,t)."""
    import tempfile
    n = len(ps)
    with tempfile.TemporaryDirectory(prefix="cp_", dir=workdir) as d:
        fi, fo = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fi, "wb") as fh:
            fh.write(struct.pack("<i", n))
            fh.write(np.ascontiguousarray(np.stack([ps, t, td, stress]).astype(np.float64)).tobytes())     # column-major a(n,4)
        exe = os.path.join(HERE, "_ref", f"cpref_{kind}")
        res = subprocess.run([exe, fi, fo], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"reference calcpar leaf driver failed: {res.stdout}\n{res.stderr}")
        return np.frombuffer(open(fo, "rb").read(), dtype=np.float64).reshape(3, n).T.copy()


def have_conv_ref(kind="r8", nest=False):
    return os.access(os.path.join(HERE, "_ref", f"convref_{kind}" + ("n" if nest else "")), os.X_OK)


def run_conv_reference(cs, kind="r8", workdir="/tmp", fm_cap=8, gpu=False):
    """The unmodified CONVECT / TLIFT, redist, sort2, f_qvsat, ew, ran3 (behind oracle/ref_conv_driver.f90, which carries our
    restatement of the calcmatrix / convmix glue) on a synthetic.convection_case() dict -> one dict per call."""
    import tempfile
    nx, ny, nuvz = (int(v) for v in cs["grid"])
    n, nl = int(cs["npart"]), int(cs["nconvlev"])
    ncalls = len(cs["itimes"])
    f8 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes()
    with tempfile.TemporaryDirectory(prefix="conv_", dir=workdir) as d:
        fi, fo = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fi, "wb") as fh:
            fh.write(struct.pack("<12i", nx, ny, nuvz, nl, int(cs["ldirect"]), int(cs["lsynctime"]), int(cs["memtime"][0]), int(cs["memtime"][1]),
                                 n, ncalls, fm_cap, 10))
            fh.write(struct.pack("<d", float(cs["height_nz"])))
            for k in ("akz", "bkz", "akm", "bkm", "ps", "tt2", "td2", "tth", "qvh", "cbaseflux", "xtra1", "ytra1", "ztra1"):
                fh.write(f8(cs[k]))           # C order [slot][level][jy][ix] == Fortran (ix,jy,level,slot)
            fh.write(np.asarray(cs["itimes"], dtype=np.int32).tobytes())
            fh.write(np.ascontiguousarray(np.asarray(cs["due"]).T.astype(np.int32)).tobytes())     # due(n,ncalls) column-major
            if "nest" in cs:
                fh.write(struct.pack("<3i", 1, int(cs["nest"][0]), int(cs["nest"][1])))
                fh.write(f8(cs["nestgeom"]))
                for k in ("psn", "tt2n", "td2n", "tthn", "qvhn", "cbasefluxn"):
                    fh.write(f8(cs[k]))
        exe = os.path.join(HERE, "_ref", f"convref_{kind}" + ("n" if "nest" in cs else ""))
        res = subprocess.run([exe, fi, fo] + (["gpu"] if gpu else []), capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"reference convection driver failed: {res.stdout}\n{res.stderr}")
        raw = open(fo, "rb").read()
    out, off = [], 0
    def take(dt, shape):
        nonlocal off
        cnt = int(np.prod(shape))
        a = np.frombuffer(raw, dtype=dt, count=cnt, offset=off).reshape(shape).copy()
        off += a.nbytes
        return a
    for _ in range(ncalls):
        z = take(np.float64, (n,)); cb = take(np.float64, (ny, nx)); lc = take(np.int32, (ny, nx)); nt = take(np.int32, (ny, nx))
        cnt = int(take(np.int32, (1,))[0]); fid = take(np.int32, (fm_cap,)); fm = take(np.float64, (fm_cap, nl, nl))
        out.append(dict(ztra1=z, cbaseflux=cb, lconv=lc, nconvtop=nt, fm_count=cnt, fm_col=fid, fmassfrac=fm))
        if "nest" in cs:
            out[-1]["cbasefluxn"] = take(np.float64, (int(cs["nest"][1]), int(cs["nest"][0])))
    return out
