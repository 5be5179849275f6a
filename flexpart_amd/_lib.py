"""ctypes binding of the C ABI declared in include/flexpart_amd.h.

The shared library is built in-tree by `__graft_entry__.build()` (hipcc,
gfx950).  There is no CPU fallback: if the library is missing or fails to
load, importing the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# FPX_LIBRARY: another build of the same library (A/B measurements of compile-time variants)
LIB_PATH = os.environ.get("FPX_LIBRARY") or os.path.join(HERE, "csrc", "libflexpart_amd.so")

FPX_MAXSPEC = 5
DEAD = -999999999
RNG_TABLE_SEQ, RNG_TABLE_COUNTER, RNG_PHILOX = 0, 1, 2


class FpxConfig(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32), ("device", C.c_int32),
        ("compute_real_bytes", C.c_int32), ("host_real_bytes", C.c_int32),
        ("max_particles", C.c_int64),
        ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("nmixz", C.c_int32),
        ("nxmax", C.c_int32), ("nymax", C.c_int32), ("nzmax", C.c_int32),
        ("dx", C.c_double), ("dy", C.c_double), ("xlon0", C.c_double), ("ylat0", C.c_double),
        ("xglobal", C.c_int32), ("nglobal", C.c_int32), ("sglobal", C.c_int32),
        ("switchnorthg", C.c_double), ("switchsouthg", C.c_double),
        ("northpolemap", C.c_double * 9), ("southpolemap", C.c_double * 9),
        ("ldirect", C.c_int32), ("lsynctime", C.c_int32), ("method", C.c_int32),
        ("mintime", C.c_int32), ("ifine", C.c_int32), ("turbswitch", C.c_int32),
        ("cblflag", C.c_int32), ("mdomainfill", C.c_int32), ("lsettling", C.c_int32),
        ("ctl", C.c_double),
        ("d_trop", C.c_double), ("d_strat", C.c_double), ("turbmesoscale", C.c_double),
        ("nspec", C.c_int32), ("maxspec", C.c_int32),
        ("drydep", C.c_int32), ("drydepspec", C.c_int32 * FPX_MAXSPEC),
        ("density", C.c_double * FPX_MAXSPEC), ("dquer", C.c_double * FPX_MAXSPEC),
        ("vsetaver", C.c_double * FPX_MAXSPEC), ("cunningham", C.c_double * FPX_MAXSPEC),
        ("decay", C.c_double * FPX_MAXSPEC),
        ("mquasilag", C.c_int32), ("lage_last", C.c_int32),
        ("rng_mode", C.c_int32), ("seed", C.c_uint64),
        ("sort_interval", C.c_int32), ("par_nxmax", C.c_int32), ("particle_base", C.c_int64),
        ("drybkdep", C.c_int32), ("wetbkdep", C.c_int32),
        ("turboff", C.c_int32), ("interpolhmix", C.c_int32), ("blend_mode", C.c_int32), ("pbl_slice_passes", C.c_int32),
        ("global_particles", C.c_int64), ("ipout", C.c_int32), ("iflux", C.c_int32), ("linit_cond", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


FPX_MAXNESTS = 4


class FpxNests(C.Structure):
    _fields_ = [("struct_bytes", C.c_int32), ("numbnests", C.c_int32), ("nxmaxn", C.c_int32), ("nymaxn", C.c_int32),
                ("nxn", C.c_int32 * FPX_MAXNESTS), ("nyn", C.c_int32 * FPX_MAXNESTS)] + \
               [(n, C.c_double * FPX_MAXNESTS) for n in ("xln", "yln", "xrn", "yrn", "xresoln", "yresoln")]


class FpxFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("uu", "vv", "ww", "uupol", "vvpol", "rho", "drhodz", "tt",
                 "hmix", "ustar", "wstar", "oli", "tropopause", "vdep")]


class FpxModelLevels(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("uuh", "vvh", "pvh", "wwh", "tth", "qvh", "ps", "tt2", "td2", "akz", "bkz", "aknew", "bknew")] + \
               [("nuvz", C.c_int32), ("nwz", C.c_int32), ("init", C.c_int32), ("pin_host", C.c_int32),
                ("nest_dy", C.c_double), ("nest_ylat0", C.c_double)]


class FpxFieldsOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("uu", "vv", "ww", "tt", "qv", "pv", "rho", "drhodz", "uupol", "vvpol", "height")] + \
               [("nmixz", C.POINTER(C.c_int32))]


class FpxCalcparIn(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("surfstr", "sshf", "akm", "bkm", "excessoro", "vdep")] + \
               [("lsubgrid", C.c_int32), ("reserved", C.c_int32 * 3)]


class FpxCalcparOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ustar", "wstar", "oli", "hmix", "tropopause")]


class FpxConvConfig(C.Structure):
    _fields_ = [("struct_bytes", C.c_int32), ("nuvz", C.c_int32), ("nconvlev", C.c_int32), ("reserved", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("akz", "bkz", "akm", "bkm")]


class FpxConvFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ps", "tt2", "td2", "tth", "qvh")] + [("nuvzmax", C.c_int32), ("reserved", C.c_int32)]


class FpxDiagFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("oro", "pv", "qv", "tt")]


class FpxRestart(C.Structure):
    _fields_ = [("jul_header", C.c_double), ("bdate", C.c_double), ("mintime", C.c_int32), ("nclassunc", C.c_int32),
                ("itrasplit", C.c_int32), ("reserved", C.c_int32)]


class FpxConcout(C.Structure):
    _fields_ = [("area", C.c_void_p), ("volume", C.c_void_p), ("outnum", C.c_double), ("wetdep", C.c_int32), ("drydep", C.c_int32),
                ("nest", C.c_int32), ("iout", C.c_int32), ("prefix_pptv", C.c_char_p), ("outheight", C.c_void_p),
                ("outlon0", C.c_double), ("outlat0", C.c_double), ("weightmolar", C.c_double * 5),
                ("reduced", C.c_int32), ("reserved", C.c_int32)]


class FpxParticles(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("xtra1", "ytra1", "ztra1", "uap", "ucp", "uzp", "us", "vs", "ws",
                 "itra1", "itramem", "idt", "npoint", "nclass", "cbt", "xmass1")] + \
               [("xmass1_ld", C.c_int64), ("itrasplit", C.c_void_p), ("xscav_frac1", C.c_void_p)]


class FpxRelease(C.Structure):
    _fields_ = [("struct_bytes", C.c_int32), ("numpoint", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("ireleasestart", "ireleaseend", "kindz", "xpoint1", "xpoint2", "ypoint1", "ypoint2",
                                          "zpoint1", "zpoint2", "point_hour", "area_hour", "point_dow", "area_dow")] + \
               [("bdate", C.c_double), ("itsplit", C.c_int32), ("ind_rel", C.c_int32), ("nclassunc", C.c_int32),
                ("reserved", C.c_int32 * 5)]


FPX_MAXAGECLASS = 8


class FpxOutgrid(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32),
        ("numxgrid", C.c_int32), ("numygrid", C.c_int32), ("numzgrid", C.c_int32),
        ("dxout", C.c_double), ("dyout", C.c_double), ("xoutshift", C.c_double), ("youtshift", C.c_double),
        ("maxpointspec_act", C.c_int32), ("nclassunc", C.c_int32), ("nageclass", C.c_int32),
        ("lage", C.c_int32 * FPX_MAXAGECLASS),
        ("ind_samp", C.c_int32), ("ioutputforeachrelease", C.c_int32), ("lusekerneloutput", C.c_int32),
        ("reserved", C.c_int32 * 5),
    ]


class FpxOutgridNest(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32),
        ("numxgridn", C.c_int32), ("numygridn", C.c_int32),
        ("dxoutn", C.c_double), ("dyoutn", C.c_double), ("xoutshiftn", C.c_double), ("youtshiftn", C.c_double),
        ("reserved", C.c_int32 * 4),
    ]


class FpxWetConfig(C.Structure):
    _fields_ = [("struct_bytes", C.c_int32), ("wetdepspec", C.c_int32 * FPX_MAXSPEC)] + \
               [(n, C.c_double * FPX_MAXSPEC) for n in ("weta_gas", "wetb_gas", "crain_aero", "csnow_aero",
                                                         "ccn_aero", "in_aero", "henry")] + \
               [("readclouds", C.c_int32), ("reserved", C.c_int32 * 7)]


class FpxWetFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("lsprec", "convprec", "tcc", "ctwc", "tt", "clouds", "cloudsh")]


class FpxStepStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("n_due", "n_initialized", "n_left_domain", "n_min_mass", "n_max_age",
                 "nan_count", "nan_count2", "n_bad_position")] + [("kernel_ms", C.c_double)]


# fpx_allreduce_fn: int fn(void *user, const void *send, void *recv, int64_t count, int32_t dtype)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)

# every symbol include/flexpart_amd.h declares (tests check the library exports all of them)
SYMBOLS = [
    "fpx_create", "fpx_destroy", "fpx_last_error", "fpx_abi_version", "fpx_polar_maps", "fpx_set_height",
    "fpx_upload_fields", "fpx_set_windtime", "fpx_rng_fill_table", "fpx_rng_set_table",
    "fpx_rng_get_table", "fpx_upload_particles", "fpx_download_particles", "fpx_set_numpart", "fpx_set_release_points", "fpx_set_release_heights", "fpx_release_init", "fpx_releaseparticles", "fpx_split_particles", "fpx_redist_plan", "fpx_redist_bytes", "fpx_redist_pack", "fpx_redist_unpack",
    "fpx_step", "fpx_step_async", "fpx_sync", "fpx_counters", "fpx_kernel_time", "fpx_kernel_times", "fpx_sort_particles",
    "fpx_seed_particles", "fpx_stream", "fpx_outgrid_init", "fpx_set_output_times", "fpx_conccalc",
    "fpx_get_grids", "fpx_comm_unique_id", "fpx_comm_init", "fpx_comm_init_host", "fpx_count_particles", "fpx_lane_stats", "fpx_set_option", "fpx_get_info", "fpx_wet_init", "fpx_upload_wet_fields",
    "fpx_wetdepo", "fpx_get_wetgrid", "fpx_nests_init", "fpx_upload_nest_fields", "fpx_math_probe",
    "fpx_outgrid_nest_init", "fpx_get_grids_nest", "fpx_receptors_init", "fpx_get_receptors", "fpx_upload_wet_nest_fields",
    "fpx_verttransform_ecmwf", "fpx_verttransform_nest", "fpx_verttransform_time", "fpx_calcpar", "fpx_calcpar_time", "fpx_upload_diag_fields", "fpx_partoutput", "fpx_partoutput_time", "fpx_readpartpositions", "fpx_concoutput",
    "fpx_checkpoint_write", "fpx_checkpoint_read",
    "fpx_conv_init", "fpx_upload_conv_fields", "fpx_convmix", "fpx_convmix_time", "fpx_get_cbaseflux", "fpx_set_cbaseflux",
    "fpx_upload_conv_nest_fields", "fpx_get_cbaseflux_nest", "fpx_set_cbaseflux_nest", "fpx_upload_diag_nest_fields",
]

_lib = None


def load():
    """Load libflexpart_amd.so (once).  Raises if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "flexpart_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.fpx_last_error.restype = C.c_char_p
    lib.fpx_stream.restype = vp
    lib.fpx_stream.argtypes = [vp]
    lib.fpx_math_probe.argtypes = [C.c_int32, vp, vp, C.c_int64]
    lib.fpx_upload_wet_nest_fields.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(FpxWetFields), C.c_int32]
    lib.fpx_outgrid_nest_init.argtypes = [vp, C.POINTER(FpxOutgridNest)]
    lib.fpx_get_grids_nest.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32]
    lib.fpx_receptors_init.argtypes = [vp, C.c_int32, vp, vp, vp]
    lib.fpx_get_receptors.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32]
    lib.fpx_create.argtypes = [C.POINTER(vp), C.POINTER(FpxConfig)]
    lib.fpx_destroy.argtypes = [vp]
    lib.fpx_polar_maps.argtypes = [C.c_int32, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.fpx_set_height.argtypes = [vp, vp, C.c_int32]
    lib.fpx_upload_fields.argtypes = [vp, C.c_int32, C.POINTER(FpxFields)]
    lib.fpx_verttransform_ecmwf.argtypes = [vp, C.c_int32, C.POINTER(FpxModelLevels), C.POINTER(FpxFields), C.POINTER(FpxFieldsOut)]
    lib.fpx_verttransform_nest.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(FpxModelLevels), C.POINTER(FpxFields), C.POINTER(FpxFieldsOut)]
    lib.fpx_verttransform_time.argtypes = [vp, C.POINTER(C.c_double)]
    lib.fpx_calcpar.argtypes = [vp, C.c_int32, C.POINTER(FpxCalcparIn), C.POINTER(FpxCalcparOut)]
    lib.fpx_calcpar_time.argtypes = [vp, C.POINTER(C.c_double)]
    lib.fpx_upload_diag_fields.argtypes = [vp, C.c_int32, C.POINTER(FpxDiagFields)]
    lib.fpx_partoutput.argtypes = [vp, C.c_int32, C.c_char_p, C.POINTER(C.c_int64)]
    lib.fpx_partoutput_time.argtypes = [vp, C.POINTER(C.c_double)]
    lib.fpx_concoutput.argtypes = [vp, C.c_int32, C.POINTER(FpxConcout), C.c_char_p, C.c_int32]
    lib.fpx_readpartpositions.argtypes = [vp, C.c_char_p, C.POINTER(FpxRestart), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.fpx_conv_init.argtypes = [vp, C.POINTER(FpxConvConfig)]
    lib.fpx_upload_conv_fields.argtypes = [vp, C.c_int32, C.POINTER(FpxConvFields)]
    lib.fpx_convmix.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64)]
    lib.fpx_convmix_time.argtypes = [vp, C.POINTER(C.c_double)]
    lib.fpx_upload_conv_nest_fields.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(FpxConvFields)]
    lib.fpx_upload_diag_nest_fields.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(FpxDiagFields)]
    lib.fpx_get_cbaseflux_nest.argtypes = [vp, C.c_int32, vp]
    lib.fpx_set_cbaseflux_nest.argtypes = [vp, C.c_int32, vp]
    lib.fpx_get_cbaseflux.argtypes = [vp, vp]
    lib.fpx_set_cbaseflux.argtypes = [vp, vp]
    lib.fpx_checkpoint_write.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32]
    lib.fpx_checkpoint_read.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.fpx_set_windtime.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.fpx_rng_fill_table.argtypes = [vp]
    lib.fpx_rng_set_table.argtypes = [vp, vp, C.c_int32]
    lib.fpx_rng_get_table.argtypes = [vp, vp, C.c_int32]
    lib.fpx_upload_particles.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(FpxParticles)]
    lib.fpx_download_particles.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(FpxParticles)]
    lib.fpx_set_numpart.argtypes = [vp, C.c_int64]
    lib.fpx_set_release_points.argtypes = [vp, C.c_int32, vp, C.POINTER(C.c_int32)]
    lib.fpx_release_init.argtypes = [vp, C.POINTER(FpxRelease)]
    lib.fpx_releaseparticles.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), vp, vp, C.POINTER(C.c_int64)]
    lib.fpx_split_particles.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64)]
    lib.fpx_redist_plan.argtypes = [C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.fpx_redist_bytes.argtypes = [vp, C.c_int64]
    lib.fpx_redist_bytes.restype = C.c_uint64
    lib.fpx_redist_pack.argtypes = [vp, C.c_int32, C.c_int64, vp, C.c_uint64, C.POINTER(C.c_int64)]
    lib.fpx_redist_unpack.argtypes = [vp, C.c_int32, C.c_int64, vp, C.c_uint64, C.POINTER(C.c_int64)]
    lib.fpx_step.argtypes = [vp, C.c_int32, C.POINTER(FpxStepStats)]
    lib.fpx_step_async.argtypes = [vp, C.c_int32]
    lib.fpx_sync.argtypes = [vp]
    lib.fpx_counters.argtypes = [vp, C.POINTER(FpxStepStats), C.c_int32]
    lib.fpx_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]
    lib.fpx_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]
    lib.fpx_sort_particles.argtypes = [vp]
    lib.fpx_seed_particles.argtypes = [vp, C.c_int64, C.c_uint64, C.c_double, C.c_double,
                                       C.c_double, C.c_int32]
    lib.fpx_outgrid_init.argtypes = [vp, C.POINTER(FpxOutgrid), vp]
    lib.fpx_set_output_times.argtypes = [vp, C.c_int32, C.c_int32]
    lib.fpx_conccalc.argtypes = [vp, C.c_int32, C.c_double]
    lib.fpx_get_grids.argtypes = [vp, vp, vp, C.c_int32, C.c_int32]
    lib.fpx_comm_unique_id.argtypes = [vp, C.c_int32]
    lib.fpx_comm_init.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32]
    lib.fpx_nests_init.argtypes = [vp, C.POINTER(FpxNests)]
    lib.fpx_upload_nest_fields.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(FpxFields)]
    lib.fpx_wet_init.argtypes = [vp, C.POINTER(FpxWetConfig)]
    lib.fpx_upload_wet_fields.argtypes = [vp, C.c_int32, C.POINTER(FpxWetFields)]
    lib.fpx_wetdepo.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    lib.fpx_get_wetgrid.argtypes = [vp, vp, C.c_int32]
    lib.fpx_comm_init_host.argtypes = [vp, C.c_int32, C.c_int32, ALLREDUCE_FN, vp]
    lib.fpx_set_release_heights.argtypes = [vp, C.c_int32, vp, vp]
    lib.fpx_count_particles.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]
    lib.fpx_lane_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int32, C.c_int32]
    lib.fpx_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.fpx_get_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    _lib = lib
    return lib


class FpxError(RuntimeError):
    def __init__(self, code, where):
        msg = load().fpx_last_error()
        super().__init__(f"{where}: status {code}: {msg.decode() if msg else ''}")
        self.code = code


def check(code, where):
    if code != 0:
        raise FpxError(code, where)
