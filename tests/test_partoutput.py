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


# ---------------------------------------------------------------------------------------------
# readpartpositions: the warm start from the dump
# ---------------------------------------------------------------------------------------------
RP_KEYS = ("xtra1", "ytra1", "ztra1", "npoint", "itramem", "nclass", "idt", "itra1", "xmass1")


def restart_setup(sc, nspec, maxpart=5000):
    # the previous run started 2020-01-01 00:00 and dumped at itime = 3600 s; this run starts 01:00:00
    return dict(geom=sc["geom"], nspec=nspec, restart=[20200101, 10000, 20200101, 0, 1, 5, 7200, 1, 1, maxpart])


@pytest.mark.parametrize("kind", ["r4", "r8"])
def test_readpart_oracle_equals_reference_fixture(kind):
    """oracle/readpart_oracle.c on the reference's own dump == the arrays the unmodified readpartpositions
    filled (tests/golden/rp_s2_<kind>.npz, made by make_golden_po.py)."""
    from oracle import oracle as orc
    sc = scenario(2)
    dump = open(os.path.join(HERE, "golden", f"po_s2_{kind}.bin"), "rb").read()
    gold = np.load(os.path.join(HERE, "golden", f"rp_s2_{kind}.npz"))
    got = orc.rp_oracle(dump, restart_setup(sc, 2), kind)
    assert got["numpart"] == int(gold["numpart"]) and got["numparticlecount"] == int(gold["numparticlecount"])
    for k in RP_KEYS + ("itrasplit",):
        assert np.array_equal(got[k], gold[k]), k


@pytest.mark.parametrize("kind", ["r4", "r8"])
def test_readpart_class_draw_equals_reference_fixture(kind):
    """nclassunc = 3: every particle of a warm start draws its uncertainty class from ran1 (seed -8), in the order of the
    records (readpartpositions.f90:142-143).  Fixture from the reference built with nclassunc = 3 (rpref_r4c / rpref_r8c)."""
    from oracle import oracle as orc
    sc = scenario(2)
    dump = open(os.path.join(HERE, "golden", f"po_s2_{kind}.bin"), "rb").read()
    gold = np.load(os.path.join(HERE, "golden", f"rp_s2_classes_{kind}.npz"))
    rs = restart_setup(sc, 2)
    rs["restart"][7] = 3
    got = orc.rp_oracle(dump, rs, kind)
    assert np.array_equal(got["nclass"], gold["nclass"]) and set(np.unique(gold["nclass"])) == {1, 2, 3}
    for k in RP_KEYS + ("itrasplit",):
        assert np.array_equal(got[k], gold[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_hip_readpartpositions_class_draw(built, tmp_path, kind):
    """The same draw on the device path: fpx_readpartpositions(nclassunc = 3) replays the serial ran1 stream on the host and
    hands every record its class; equal to the fixture of the reference built with nclassunc = 3."""
    from flexpart_amd.engine import Engine, RNG_PHILOX
    from oracle import oracle as orc
    sc = scenario(2)
    rb = 8 if kind == "r8" else 4
    path = tmp_path / "partposit_end"
    path.write_bytes(open(os.path.join(HERE, "golden", f"po_s2_{kind}.bin"), "rb").read())
    gold = np.load(os.path.join(HERE, "golden", f"rp_s2_classes_{kind}.npz"))
    sc2 = {k: v for k, v in sc.items() if k not in ("npart", "xtra1", "ytra1", "ztra1", "itra1", "itramem", "npoint", "nclass", "idt", "uap", "ucp", "uzp", "us", "vs", "ws", "cbt", "xmass1")}
    b = Engine(sc2, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=RNG_PHILOX, max_particles=4000)
    n, npc, itimein = b.readpartpositions(path, orc.juldate(20200101, 0, kind), orc.juldate(20200101, 10000, kind), mintime=5, nclassunc=3)
    got = b.download()
    b.close()
    assert n == int(gold["numpart"])
    for k in RP_KEYS:
        assert np.array_equal(got[k], gold[k]), k


@pytest.mark.ref
@pytest.mark.parametrize("kind", ["r4", "r8"])
def test_readpart_oracle_equals_live_reference(kind):
    from oracle import oracle as orc, scenario_io as sio
    if not sio.have_rp_ref(kind):
        pytest.skip("flang-built reference not present (GPU box)")
    sc = scenario(3, n=700, seed=9)
    dump = orc.po_oracle(sc, kind)
    rs = restart_setup(sc, 3)
    ref = sio.run_rp_reference(dump, rs, kind)
    got = orc.rp_oracle(dump, rs, kind)
    assert got["numpart"] == ref["numpart"]
    for k in RP_KEYS + ("itrasplit",):
        assert np.array_equal(got[k], ref[k]), k


def test_readpart_rejects_a_run_that_does_not_continue():
    from oracle import oracle as orc
    sc = scenario(1)
    rs = restart_setup(sc, 1)
    rs["restart"][1] = 20000          # this run starts at 02:00, the dump is from 01:00
    assert orc.rp_oracle(orc.po_oracle(sc, "r8"), rs, "r8")["status"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_hip_readpartpositions_matches_oracle(built, tmp_path, kind):
    """Dump written by the device, read back by the device into a fresh engine: every array equals what the
    oracle's readpartpositions makes of the same file (bit-exact: integer work plus one subtraction and division)."""
    from flexpart_amd.engine import Engine, RNG_PHILOX
    from oracle import oracle as orc
    sc = scenario(2)
    rb = 8 if kind == "r8" else 4
    a = Engine(sc, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=RNG_PHILOX)
    a.upload_diag_fields_from_scenario(sc)
    a.sort()
    path = tmp_path / "partposit_end"
    nrec = a.partoutput(3600, path)
    a.close()
    rs = restart_setup(sc, 2)
    want = orc.rp_oracle(path.read_bytes(), rs, kind)
    sc2 = {k: v for k, v in sc.items() if k not in ("npart", "xtra1", "ytra1", "ztra1", "itra1", "itramem", "npoint", "nclass", "idt", "uap", "ucp", "uzp", "us", "vs", "ws", "cbt", "xmass1")}
    b = Engine(sc2, compute_real_bytes=rb, host_real_bytes=rb, rng_mode=RNG_PHILOX, max_particles=4000)
    jul = orc.juldate(20200101, 0, kind)
    n, npc, itimein = b.readpartpositions(path, jul, orc.juldate(20200101, 10000, kind), mintime=5)
    got = b.download()
    assert (n, npc, itimein) == (nrec, want["numparticlecount"], 3600)
    for k in RP_KEYS:
        assert np.array_equal(got[k], want[k]), k
    assert not got["uap"].any() and (got["cbt"] == 1).all()
    # a dump that does not continue this run is refused (readpartpositions.f90:134)
    with pytest.raises(Exception):
        b.readpartpositions(path, jul, orc.juldate(20200101, 20000, kind), mintime=5)
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_fortran_host_readpartpositions(built, kind):
    """The real Fortran host: oracle/_ref/rpref_rK either calls the reference's readpartpositions or lets the
    engine parse the same dump (flexgpu_readpartpositions) and takes the arrays back into com_mod."""
    from oracle import oracle as orc, scenario_io as sio
    if not sio.have_rp_ref(kind):
        pytest.skip("oracle/_ref binaries not present in this snapshot")
    sc = scenario(2, n=800, seed=3)
    dump = orc.po_oracle(sc, kind)
    rs = restart_setup(sc, 2)
    ref = sio.run_rp_reference(dump, rs, kind)
    gpu = sio.run_rp_reference(dump, rs, kind, gpu=True)
    assert gpu["numpart"] == ref["numpart"] and gpu["numparticlecount"] == ref["numparticlecount"]
    for k in RP_KEYS + ("itrasplit",):
        assert np.array_equal(gpu[k], ref[k]), k


@pytest.mark.gpu
def test_hip_partoutput_large_dump_properties(built, tmp_path):
    """5e6 particles seeded on the device on the BASELINE grid (no host copy of them exists): the size-independent
    properties of the dump -- file size from the number due, every record framed by its length, the closing record,
    finite heights, and a second dump after a locality sort is byte-identical: the records are in particle-number
    order whatever the device order is."""
    import hashlib
    from flexpart_amd.engine import Engine, RNG_PHILOX
    sc = syn.base_scenario(ctl=5.0, ifine=4, nsteps=1)
    sc["npart"] = 1
    syn.add_partoutput_fields(sc, itime=0, dead_every=0)
    del sc["npart"], sc["itra1"], sc["npoint"]
    n = 5_000_000
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=8, rng_mode=RNG_PHILOX, max_particles=n, sort_interval=4)
    eng.upload_diag_fields_from_scenario(sc)
    eng.seed_particles(n, seed=0x5EED, frac_pbl=0.5)
    p1, p2 = tmp_path / "a", tmp_path / "b"
    assert eng.partoutput(0, p1) == n
    eng.sort()
    assert eng.partoutput(0, p2) == n
    eng.close()
    rl = 8 + 11 * 8
    assert p1.stat().st_size == 12 + (n + 1) * (rl + 8)
    h = [hashlib.sha256(p.read_bytes()).hexdigest() for p in (p1, p2)]
    assert h[0] == h[1]
    body = np.memmap(p1, np.uint8, mode="r", offset=12).reshape(n + 1, rl + 8)
    assert (body[:, :4].view(np.int32) == rl).all() and (body[:, -4:].view(np.int32) == rl).all()
    assert body[-1, 4:8].view(np.int32)[0] == -99999
    z = body[:-1, 4 + 4 + 16:4 + 4 + 24].copy().view(np.float64)[:, 0]
    assert np.isfinite(z).all() and z.min() >= 0.0
