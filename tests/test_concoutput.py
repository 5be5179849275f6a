"""concoutput: the sparse concentration files grid_conc_<date><time>_<species>.

CPU (-m "not gpu"): the C restatement oracle/concoutput_oracle.c reproduces byte for byte the files the
unmodified reference routine wrote (tests/golden/co_*.bin, made by tests/golden/make_golden_co.py with the
flang build; float only -- concoutput.f90 does not compile with -fdefault-real-8) and the live reference.
GPU (-m gpu): particles are sampled into the output grid by the device (conccalc, dry and wet deposition),
fpx_concoutput compresses those grids on the device and writes the files; the oracle gets the very same grids
(downloaded through fpx_get_grids) and must produce the same bytes.
"""
import os

import numpy as np
import pytest

from flexpart_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"two_species": dict(), "no_deposition": dict(wet=False, dry=False, nspec=1), "odd_sizes": dict(nxg=37, nyg=19, nzg=3, nspec=3, seed=11),
         # three uncertainty classes: the files hold the class mean times nclassunc (concoutput.f90:296-345, mean_mod.f90); the
         # fixture comes from the reference built with nclassunc = 3 (coref_r4c, oracle/build_ref.sh)
         "classes": dict(nxg=30, nyg=20, nzg=3, nspec=2, seed=31, classes=3)}


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_equals_reference_files(case):
    from oracle import oracle as orc
    got = orc.co_oracle(syn.concoutput_case(**CASES[case]))
    for suffix, b in got.items():
        assert b == open(os.path.join(HERE, "golden", f"co_{case}{suffix}.bin"), "rb").read(), suffix


@pytest.mark.ref
def test_oracle_equals_live_reference():
    from oracle import oracle as orc, scenario_io as sio
    if not sio.have_co_ref():
        pytest.skip("flang-built reference not present (GPU box)")
    co = syn.concoutput_case(nxg=50, nyg=31, nzg=5, nspec=2, seed=99)
    ref = sio.run_co_reference(co)
    got = orc.co_oracle(co)
    assert len(ref) == 2
    for name, b in ref.items():
        assert got["_" + name[-3:]] == b, name


def test_runs_and_signs():
    """The compression itself on a hand-made row: runs start where a non-zero cell follows a zero one, the
    sign of the values flips from run to run, indices of the 3-D dump are offset by one level."""
    import struct
    from oracle import oracle as orc
    co = syn.concoutput_case(nxg=8, nyg=4, nzg=2, nspec=1, wet=False, dry=False)
    co["gridunc"] = np.zeros((1, 2, 4, 8))
    co["gridunc"][0, 0, 0, :] = np.array([0, 1, 2, 0, 0, 3, 0, 4]) * 1e-3
    b = orc.co_oracle(co)["_001"]
    p = 12 + 2 * (12 + 8 + 12 + 8)              # itime record, empty wet and dry dumps
    ci = struct.unpack("<i", b[p + 4:p + 8])[0]
    idx = struct.unpack(f"<{ci}i", b[p + 16:p + 16 + 4 * ci])
    q = p + 12 + 8 + 4 * ci
    cr = struct.unpack("<i", b[q + 4:q + 8])[0]
    vals = np.frombuffer(b[q + 16:q + 16 + 4 * cr], np.float32)
    assert (ci, cr) == (3, 4) and idx == (32 + 1, 32 + 5, 32 + 7)      # + numxgrid*numygrid: kz is 1-based in the index
    assert list(np.sign(vals)) == [1, 1, -1, 1]


@pytest.mark.ref
def test_oracle_class_mean_equals_live_reference():
    from oracle import oracle as orc, scenario_io as sio
    if not os.access(os.path.join(os.path.dirname(sio.__file__), "_ref", "coref_r4c"), os.X_OK):
        pytest.skip("flang-built reference (nclassunc = 3) not present (GPU box)")
    co = syn.concoutput_case(nxg=41, nyg=23, nzg=4, nspec=3, seed=77, classes=3)
    ref = sio.run_co_reference(co, kind="r4c")
    got = orc.co_oracle(co)
    assert len(ref) == 3
    for name, b in ref.items():
        assert got["_" + name[-3:]] == b, name


@pytest.mark.gpu
@pytest.mark.parametrize("compute", [8, 4])
def test_hip_concoutput_class_mean(built, tmp_path, compute):
    """nclassunc = 3: the device sums the three class planes of gridunc / wetgridunc / drygridunc as mean_mod does
    (mean times nclassunc, concoutput.f90:323-345) before the run-length compression; the files equal the oracle's, which
    is pinned byte for byte to the reference built with nclassunc = 3 (tests/golden/co_classes_*.bin)."""
    from flexpart_amd.engine import Engine
    from oracle import oracle as orc
    from test_oracle_cpu import golden_scenario
    sc = syn.add_wet(syn.add_outgrid(golden_scenario("aerosol")))
    syn.add_release_points(sc, xmass=[[1.0]], npart_rel=[int(sc["npart"])], ioutputforeachrelease=0, nclassunc=3)
    eng = Engine(sc, compute_real_bytes=compute, host_real_bytes=4)
    eng.run()
    g, d = eng.grids()
    w = eng.wetgrid()
    na, nc, mp, nsp, nzg, nyg, nxg = eng.gshape
    assert (na, nc, mp) == (1, 3, 1) and all((g[0, c] > 0).sum() > 100 for c in range(3))
    case = syn.concoutput_case(nxg=nxg, nyg=nyg, nzg=nzg, nspec=nsp)
    prefix = str(tmp_path / "grid_conc_20200101010000_")
    eng.concoutput(3600, prefix, case["area"], case["volume"], outnum=4.0, wetdep=True, drydep=True, clear=True)
    eng.close()
    co = dict(outgrid=np.array([nxg, nyg, nzg, nsp, 1, 1, 3600], np.int32), outgeom=case["outgeom"], outheight=case["outheight"],
              area=case["area"], volume=case["volume"], classes=np.array([3], np.int32),
              gridunc=g[0, :, 0], wetgridunc=w[0, :, 0], drygridunc=d[0, :, 0])
    want = orc.co_oracle(co)
    for suffix, b in want.items():
        assert open(prefix + suffix[1:], "rb").read() == b, suffix


@pytest.mark.gpu
@pytest.mark.parametrize("compute", [8, 4])
def test_hip_concoutput_files_are_byte_identical(built, tmp_path, compute):
    from flexpart_amd.engine import Engine
    from oracle import oracle as orc
    from test_oracle_cpu import golden_scenario
    sc = syn.add_wet(syn.add_outgrid(golden_scenario("aerosol")))
    eng = Engine(sc, compute_real_bytes=compute, host_real_bytes=4)
    eng.run()
    g, d = eng.grids()
    w = eng.wetgrid()
    na, nc, mp, nsp, nzg, nyg, nxg = eng.gshape
    assert (na, nc, mp) == (1, 1, 1) and (g > 0).sum() > 300 and (d > 0).sum() > 50
    case = syn.concoutput_case(nxg=nxg, nyg=nyg, nzg=nzg, nspec=nsp)
    area, volume = case["area"], case["volume"]
    prefix = str(tmp_path / "grid_conc_20200101010000_")
    eng.concoutput(3600, prefix, area, volume, outnum=4.0, wetdep=True, drydep=True, clear=True)
    g2, _ = eng.grids()
    eng.close()
    assert not g2.any()                                   # gridunc is zeroed after the output, concoutput.f90:714
    co = dict(outgrid=np.array([nxg, nyg, nzg, nsp, 1, 1, 3600], np.int32), outgeom=case["outgeom"], outheight=case["outheight"],
              area=area, volume=volume, gridunc=g[0, 0, 0], wetgridunc=w[0, 0, 0], drygridunc=d[0, 0, 0])
    want = orc.co_oracle(co)
    for suffix, b in want.items():
        assert open(prefix + suffix[1:], "rb").read() == b, suffix


def test_oracle_equals_reference_files_of_concoutput_nest():
    """The nested output grid: concoutput_nest.f90 is the same algorithm on griduncn / arean / volumen; the fixture is
    the file the unmodified routine wrote (tests/golden/co_nest_*.bin)."""
    from oracle import oracle as orc
    got = orc.co_oracle(syn.concoutput_case(nxg=30, nyg=20, nzg=3, nspec=2, seed=21))
    for suffix, b in got.items():
        assert b == open(os.path.join(HERE, "golden", f"co_nest{suffix}.bin"), "rb").read(), suffix


@pytest.mark.gpu
def test_hip_concoutput_nested_output_grid(built, tmp_path):
    """Mother and nested output grid sampled by the device; fpx_concoutput(nest=1) against the oracle on the
    downloaded griduncn / drygriduncn / wetgriduncn."""
    from flexpart_amd.engine import Engine
    from oracle import oracle as orc
    from test_oracle_cpu import golden_scenario
    sc = golden_scenario("sampling_nest")
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=4)
    eng.run()
    g, d, w = eng.grids_nest()
    na, nc, mp, nsp, nzg, nyn, nxn = eng.gshape_nest
    assert (g > 0).sum() > 100
    case = syn.concoutput_case(nxg=nxn, nyg=nyn, nzg=nzg, nspec=nsp)
    prefix = str(tmp_path / "grid_conc_nest_20200101010000_")
    wet = bool(sc.get("wetdep", 0)); dry = bool(sc.get("drydep", 0))
    eng.concoutput(3600, prefix, case["area"], case["volume"], outnum=2.0, wetdep=wet, drydep=dry, nest=True)
    eng.close()
    geom = case["outgeom"].copy(); geom[4] = 2.0
    co = dict(outgrid=np.array([nxn, nyn, nzg, nsp, int(wet), int(dry), 3600], np.int32), outgeom=geom, outheight=case["outheight"],
              area=case["area"], volume=case["volume"], gridunc=g[0, 0, 0], wetgridunc=w[0, 0, 0], drygridunc=d[0, 0, 0])
    for suffix, b in orc.co_oracle(co).items():
        assert open(prefix + suffix[1:], "rb").read() == b, suffix


def test_oracle_equals_reference_pptv_files():
    """iout = 3: grid_conc_* and grid_pptv_* (mixing ratio: densityoutgrid from the met density of slot memind(2), molar
    weights) against the files the unmodified routine wrote (tests/golden/co_pptv_*.bin)."""
    from oracle import oracle as orc
    got = orc.co_oracle(syn.add_pptv(syn.concoutput_case(nxg=30, nyg=20, nzg=4, nspec=2, seed=8)))
    assert sorted(got) == ["_001", "_002", "pptv_001", "pptv_002"]
    for key, b in got.items():
        assert b == open(os.path.join(HERE, "golden", f"co_pptv_{key.strip('_')}.bin"), "rb").read(), key


@pytest.mark.gpu
def test_hip_concoutput_mixing_ratio_files(built, tmp_path):
    """iout = 3 on the device: the air density of the output cells comes from the device's own met pack."""
    from flexpart_amd.engine import Engine
    from oracle import oracle as orc
    from test_oracle_cpu import golden_scenario
    sc = syn.add_outgrid(golden_scenario("aerosol"))
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=4)
    eng.run()
    g, d = eng.grids()
    na, nc, mp, nsp, nzg, nyg, nxg = eng.gshape
    dxo, dyo, lon0, lat0 = (float(v) for v in sc["outgeom"])
    case = syn.concoutput_case(nxg=nxg, nyg=nyg, nzg=nzg, nspec=nsp)
    outheight = np.asarray(sc["outheight"], np.float64)
    wm = [350.0]
    pc, pp = str(tmp_path / "grid_conc_x_"), str(tmp_path / "grid_pptv_x_")
    eng.concoutput(3600, pc, case["area"], case["volume"], outnum=4.0, drydep=True, iout=3, prefix_pptv=pp, outheight=outheight,
                   outlon0=lon0, outlat0=lat0, weightmolar=wm)
    eng.close()
    m2 = int(sc["memind"][1]) - 1
    co = dict(outgrid=np.array([nxg, nyg, nzg, nsp, 0, 1, 3600], np.int32), outgeom=np.array([dxo, dyo, lon0, lat0, 4.0]), outheight=outheight,
              area=case["area"], volume=case["volume"], gridunc=g[0, 0, 0], drygridunc=d[0, 0, 0],
              iout=3, met=sc["grid"], metgeom=sc["geom"], height=sc["height"], rho2=np.asarray(sc["rho"])[m2], weightmolar=np.array(wm))
    want = orc.co_oracle(co)
    assert open(pc + "001", "rb").read() == want["_001"]
    assert open(pp + "001", "rb").read() == want["pptv_001"]


@pytest.mark.gpu
def test_hip_concoutput_nested_mixing_ratio(built, tmp_path):
    """grid_pptv_nest_*: iout = 2 on the nested output grid (concoutput_nest.f90 with outlon0n, dxoutn ...)."""
    from flexpart_amd.engine import Engine
    from oracle import oracle as orc
    from test_oracle_cpu import golden_scenario
    sc = golden_scenario("sampling_nest")
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=4)
    eng.run()
    g, d, w = eng.grids_nest()
    na, nc, mp, nsp, nzg, nyn, nxn = eng.gshape_nest
    dxn, dyn, lon0n, lat0n = (float(v) for v in sc["outgeomn"])
    case = syn.concoutput_case(nxg=nxn, nyg=nyn, nzg=nzg, nspec=nsp)
    outheight = np.asarray(sc["outheight"], np.float64)
    pp = str(tmp_path / "grid_pptv_nest_x_")
    eng.concoutput(3600, str(tmp_path / "unused_"), case["area"], case["volume"], outnum=2.0, nest=True, iout=2, prefix_pptv=pp,
                   outheight=outheight, outlon0=lon0n, outlat0=lat0n, weightmolar=[120.0] * nsp)
    eng.close()
    m2 = int(sc["memind"][1]) - 1
    co = dict(outgrid=np.array([nxn, nyn, nzg, nsp, 0, 0, 3600], np.int32), outgeom=np.array([dxn, dyn, lon0n, lat0n, 2.0]), outheight=outheight,
              area=case["area"], volume=case["volume"], gridunc=g[0, 0, 0],
              iout=3, met=sc["grid"], metgeom=sc["geom"], height=sc["height"], rho2=np.asarray(sc["rho"])[m2], weightmolar=np.array([120.0] * nsp))
    want = orc.co_oracle(co)
    for ks in range(nsp):
        assert open(pp + f"{ks + 1:03d}", "rb").read() == want[f"pptv_{ks + 1:03d}"]
