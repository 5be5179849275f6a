"""CPU: the C-ABI library loads without a GPU and exports every symbol the header declares."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "flexpart_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fpx_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    from flexpart_amd import _lib
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"libflexpart_amd.so does not export {n}"
    assert sorted(_lib.SYMBOLS) == names


def test_struct_sizes_match_header(built):
    """fpx_create checks struct_bytes; a mismatch between the ctypes mirror and the C header
    would make every call fail, so pin it here without needing a device."""
    from flexpart_amd import _lib
    lib = _lib.load()
    cfg = _lib.FpxConfig()
    cfg.struct_bytes = C.sizeof(_lib.FpxConfig) + 8
    h = C.c_void_p()
    rc = lib.fpx_create(C.byref(h), C.byref(cfg))
    assert rc == -1 and b"size mismatch" in lib.fpx_last_error()
    cfg.struct_bytes = C.sizeof(_lib.FpxConfig)
    cfg.host_real_bytes = 3
    rc = lib.fpx_create(C.byref(h), C.byref(cfg))
    assert rc == -1 and b"host_real_bytes" in lib.fpx_last_error()
    assert lib.fpx_abi_version() == 4


def test_no_device_is_an_error_not_a_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from flexpart_amd import _lib
    lib = _lib.load()
    cfg = _lib.FpxConfig()
    cfg.struct_bytes = C.sizeof(_lib.FpxConfig)
    cfg.host_real_bytes = 8
    cfg.compute_real_bytes = 8
    h = C.c_void_p()
    rc = lib.fpx_create(C.byref(h), C.byref(cfg))
    assert rc == -2 and not h.value


def test_null_handle_is_rejected(built):
    from flexpart_amd import _lib
    lib = _lib.load()
    assert lib.fpx_sync(None) == -1
    assert lib.fpx_step(None, 0, None) == -1


def test_ctypes_mirror_matches_the_c_header(tmp_path):
    """include/flexpart_amd.h is valid C99 and every struct the Python mirror marshals has the size the C compiler gives it."""
    import subprocess
    from flexpart_amd import _lib
    pairs = [("fpx_config", _lib.FpxConfig), ("fpx_fields", _lib.FpxFields), ("fpx_particles", _lib.FpxParticles),
             ("fpx_step_stats", _lib.FpxStepStats), ("fpx_model_levels", _lib.FpxModelLevels), ("fpx_fields_out", _lib.FpxFieldsOut),
             ("fpx_diag_fields", _lib.FpxDiagFields), ("fpx_restart", _lib.FpxRestart), ("fpx_concout", _lib.FpxConcout),
             ("fpx_nests", _lib.FpxNests), ("fpx_outgrid", _lib.FpxOutgrid), ("fpx_wet_config", _lib.FpxWetConfig),
             ("fpx_wet_fields", _lib.FpxWetFields)]
    src = tmp_path / "sizes.c"
    src.write_text('#include "flexpart_amd.h"\n#include <stdio.h>\nint main(void) {\n'
                   + "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n, _ in pairs) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n, t in pairs:
        assert int(got[n]) == C.sizeof(t), (n, got[n], C.sizeof(t))


@pytest.mark.ref
@pytest.mark.parametrize("kind", ["r4", "r8"])
def test_fortran_bind_c_types_match_the_c_header(kind):
    """flexpart_amd/fortran/flexgpu_mod.f90: c_sizeof of every bind(C) type (printed by the flang-built driver) equals
    the ctypes mirror's size, which the test above ties to the C header."""
    import subprocess
    from flexpart_amd import _lib
    exe = os.path.join(ROOT, "oracle", "_ref", f"vtref_{kind}")
    if not os.access(exe, os.X_OK):
        pytest.skip("flang-built drivers not present")
    got = [int(v) for v in subprocess.check_output([exe, "abi"], text=True).split()]
    want = [C.sizeof(t) for t in (_lib.FpxConfig, _lib.FpxFields, _lib.FpxParticles, _lib.FpxStepStats, _lib.FpxModelLevels,
                                  _lib.FpxFieldsOut, _lib.FpxDiagFields, _lib.FpxRestart, _lib.FpxConcout, _lib.FpxNests)]
    assert got == want
