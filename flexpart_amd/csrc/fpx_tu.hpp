// fpx_tu.hpp -- how the library is split into translation units.
//
// fpx_engine.hip is compiled six times, in parallel (__graft_entry__.build):
//   -DFPX_TU_REAL=8|4   the arithmetic type of the unit: 8 = the fp64 engine (and, in part 0, the C ABI),
//                        4 = the reference-typed fp32 engine (reached through fpx::make_engine_f32);
//   -DFPX_TU_PART=0|1|2 0 = the engine, every kernel except the three of the step; 1 = k_prep (16 variants);
//                        2 = k_pbl_loop + k_pbl_finish.  Parts 1 and 2 export their kernels as tables of host-stub
//                        addresses (fpx::step_kernel_*), which is how the engine launches them anyway.
// Everything inside namespace fpx lives in an inline namespace named after the unit, so the objects share no symbol
// except the ones declared outside FPX_TU_OPEN .. FPX_TU_CLOSE (the error string, EngineBase, the factory and the
// kernel tables).  No flags (FPX_TU_REAL 0, FPX_TU_PART -1): one unit with everything, `hipcc -c fpx_engine.hip`.
#pragma once
#ifndef FPX_TU_REAL
#define FPX_TU_REAL 0
#endif
#ifndef FPX_TU_PART
#define FPX_TU_PART -1
#endif
#define FPX_TU_CAT2(a, b, c) a##b##_p##c
#define FPX_TU_CAT(a, b, c) FPX_TU_CAT2(a, b, c)
#if FPX_TU_REAL != 0 && FPX_TU_REAL != 4 && FPX_TU_REAL != 8
#error "FPX_TU_REAL must be 0, 4 or 8"
#endif
#if FPX_TU_PART < 0
#define FPX_TU_OPEN inline namespace FPX_TU_CAT(tu_r, FPX_TU_REAL, all) {
#else
#define FPX_TU_OPEN inline namespace FPX_TU_CAT(tu_r, FPX_TU_REAL, FPX_TU_PART) {
#endif
#define FPX_TU_CLOSE }
