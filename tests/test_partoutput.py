"""partoutput (the binary particle dump partposit_*).

CPU (-m "not gpu"): the C restatement oracle/partoutput_oracle.c reproduces byte for byte the files the
unmodified reference routine wrote (tests/golden/po_*.bin, made by tests/golden/make_golden_po.py with
the flang build) and, where oracle/_ref exists, the live reference.
GPU (-m gpu): the file fpx_partoutput writes through the C ABI is byte-identical to the oracle's --
integer/byte work: bit-exact is the bar, also after a locality sort has permuted the device arrays.
"""
import os

import numpy as np
import pytest

from flexpart_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))


def scenario(nspec=1, n=1500, seed=1234):
    sc = syn.small(n=n, nx=40, ny=24, nz=30, nsteps=1, nspec=nspec, seed=seed)
    sc["xmass1"] = (1.0 + 0.001 * np.arange(nspec * n, dtype=np.float64)).reshape(nspec, n)
    sc["itramem"] = (-(np.arange(n) % 5) * 900).astype(np.int32)
    return syn.add_partoutput_fields(sc, itime=3600)


@pytest.mark.parametrize("kind", ["r4", "r8"])
@pytest.mark.parametrize("nspec", [1, 2])
def test_oracle_equals_reference_file(kind, nspec):
    from oracle import oracle as orc
    gold = open(os.path.join(HERE, "golden", f"po_s{nspec}_{kind}.bin"), "rb").read()
    assert orc.po_oracle(scenario(nspec), kind) == gold


@pytest.mark.ref
@pytest.mark.parametrize("kind", ["r4", "r8"])
def test_oracle_equals_live_reference(kind):
    from oracle import oracle as orc, scenario_io as sio
    if not sio.have_po_ref(kind):
        pytest.skip("flang-built reference not present (GPU box)")
    sc = scenario(3, n=900, seed=77)
    assert orc.po_oracle(sc, kind) == sio.run_po_reference(sc, kind)


def test_record_stream_is_well_formed():
    """Size-independent properties of the format: header, record markers, count, closing record."""
    import struct
    from oracle import oracle as orc
    sc = scenario(2)
    b = orc.po_oracle(sc, "r4")
    assert struct.unpack("<iii", b[:12]) == (4, 3600, 4)
    rl = 8 + 12 * 4
    body = b[12:]
    assert len(body) % (rl + 8) == 0
    nrec = len(body) // (rl + 8)
    assert nrec == int((sc["itra1"] == 3600).sum()) + 1
    recs = np.frombuffer(body, np.uint8).reshape(nrec, rl + 8)
    assert (recs[:, :4].view(np.int32) == rl).all() and (recs[:, -4:].view(np.int32) == rl).all()
    assert recs[-1, 4:8].view(np.int32)[0] == -99999


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["r8", "r4"])
@pytest.mark.parametrize("nspec,sort", [(1, False), (2, True)])
def test_hip_partoutput_file_is_byte_identical(built, tmp_path, kind, nspec, sort):
    from flexpart_amd.engine import Engine, RNG_PHILOX
    from oracle import oracle as orc
    sc = scenario(nspec)
    rb = 8 if kind == "r8" else 4
    eng = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=RNG_PHILOX, pad=(2, 3, 1))
    eng.upload_diag_fields_from_scenario(sc)
    if sort:
        eng.sort()
    path = tmp_path / "partposit_end"
    nrec = eng.partoutput(3600, path)
    eng.close()
    assert nrec == int((sc["itra1"] == 3600).sum())
    want = orc.po_oracle(sc, kind, nymax=sc["grid"][1] + 3)
    assert path.read_bytes() == want


@pytest.mark.gpu
def test_hip_partoutput_mixed_precision_host(built, tmp_path):
    """The reference as shipped (f32 host arrays) driving the fp64 engine: the dump is still the f32 file."""
    from flexpart_amd.engine import Engine, RNG_PHILOX
    from oracle import oracle as orc
    sc = scenario(1)
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=4, rng_mode=RNG_PHILOX)
    eng.upload_diag_fields_from_scenario(sc)
    path = tmp_path / "partposit_end"
    eng.partoutput(3600, path)
    eng.close()
    assert path.read_bytes() == orc.po_oracle(sc, "r4", nymax=sc["grid"][1])


@pytest.mark.gpu
def test_hip_partoutput_against_reference_file(built, tmp_path):
    """HIP path directly against the file the unmodified reference wrote (tests/golden)."""
    from flexpart_amd.engine import Engine, RNG_PHILOX
    sc = scenario(2)
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_PHILOX)
    eng.upload_diag_fields_from_scenario(sc)
    path = tmp_path / "partposit_end"
    eng.partoutput(3600, path)
    eng.close()
    assert path.read_bytes() == open(os.path.join(HERE, "golden", "po_s2_r8.bin"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_fortran_host_partoutput(built, kind):
    """The real Fortran host: oracle/_ref/poref_rK holds the reference's com_mod arrays and either calls the
    reference's partoutput or hands the same arrays to the engine (flexgpu_upload_fields / _diag_fields /
    _particles, ISO_C_BINDING) and lets flexgpu_partoutput write the file: same name, same bytes."""
    from oracle import scenario_io as sio
    if not sio.have_po_ref(kind):
        pytest.skip("oracle/_ref binaries not present in this snapshot")
    sc = scenario(2, n=1200, seed=5)
    assert sio.run_po_reference(sc, kind, gpu=True) == sio.run_po_reference(sc, kind)
