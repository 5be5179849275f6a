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


def test_kernels_that_read_their_arguments_in_place_have_them_first():
    """k_prep, k_pbl_loop, k_pbl_finish and the scatter kernels read the View (and the aggregates behind it) through the
    kernel-argument segment at offset 0 (FPX_VIEW_FROM_KERNARG / the KArgs mirror structs).  That is only right while the
    parameter list starts with exactly the members of the mirror, in the same order."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flexpart_amd", "csrc", "fpx_engine.hip")).read()
    kernels = list(re.finditer(r"__global__\s+void\s+(?:__launch_bounds__\([^;{]*?\)\s*)?(k_\w+)\(([^{;]*?)\)\s*\{", src, re.S))
    assert len(kernels) > 30
    seen = 0
    for i, m in enumerate(kernels):
        end = kernels[i + 1].start() if i + 1 < len(kernels) else len(src)
        body = src[m.end():end]
        head = body[:1500]
        if "FPX_VIEW_FROM_KERNARG(V, V_arg)" not in head and "__builtin_amdgcn_kernarg_segment_ptr()" not in head:
            continue
        seen += 1
        params = [p.strip() for p in re.sub(r"\s+", " ", m.group(2)).split(",")]
        assert params[0] == "View<R> V_arg", (m.group(1), params[0])
        mirror = re.search(r"struct KArgs \{([^}]*)\}", head)
        if mirror:
            members = [re.sub(r"\s+", " ", x.strip()) for x in mirror.group(1).split(";") if x.strip()]
            types = [x.rsplit(" ", 1)[0] for x in members]
            got = [p.rsplit(" ", 1)[0] for p in params[:len(types)]]
            assert got == types, (m.group(1), got, types)
    assert seen >= 6
