/*
 * flexpart_amd.h -- C ABI of the MI355X particle-advection engine.
 *
 * Drop-in boundary for ONE path of MeteoSwiss/flexpart: the per-particle loop
 * of the time manager (reference src/timemanager.f90:531-712, MPI twin
 * src/timemanager_mpi.f90:674-856), i.e. for every particle that is due:
 *     initialize()  (src/initialize.f90:4)   when newly released
 *     advance()     (src/advance.f90:4)      one lsynctime of motion
 *     epilogue      (src/timemanager.f90:630-708: reschedule / terminate,
 *                    decay and dry-deposition mass split)
 * plus the concentration sampling loop conccalc() (src/conccalc.f90:4: mother and nested
 * output grid, receptor points), the dry-deposition kernels and the wet-deposition loop
 * wetdepo() (src/wetdepo.f90:4) with their nested-grid variants.
 *
 * The reference has no FFI for this path: state is passed implicitly through
 * Fortran modules (com_mod, par_mod, interpol_mod, hanna_mod).  Every entry
 * point below therefore names the module variables it replaces.  All entry
 * points are extern "C", take plain pointers and sizes, return 0 on success or
 * a negative fpx_status, never throw and never call exit/stop.  A handle owns
 * all device memory; the caller owns every host array and may reuse it as soon
 * as a call returns.  Calls on one handle must be serialised by the caller
 * (the reference's time manager is single-threaded); use one handle per GPU.
 *
 * Host arrays keep the reference's conventions: column-major, x fastest,
 * 0-based x/y extents allocated with stride nxmax/nymax (NOT nx/ny), 1-based
 * levels with stride nzmax, species-major xmass1(maxpart, maxspec).
 * `host_real_bytes` states the caller's default real kind: 4 for the reference
 * as its makefile builds it, 8 for a -fdefault-real-8 build.  xtra1/ytra1 are
 * always 8-byte reals (com_mod.f90:680), cbt is integer(kind=2) (com_mod.f90:695).
 */
#ifndef FLEXPART_AMD_H
#define FLEXPART_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPX_MAXSPEC 5          /* par_mod.f90:211 maxspec  */
#define FPX_MAXNESTS 4
#define FPX_DEAD (-999999999)  /* itra1 of a terminated particle, FLEXPART.f90:315-317 */

typedef struct fpx_engine *fpx_handle;

typedef enum {
  FPX_OK = 0,
  FPX_ERR_ARG = -1,        /* bad argument / inconsistent sizes            */
  FPX_ERR_DEVICE = -2,     /* HIP runtime error (message via fpx_last_error) */
  FPX_ERR_STATE = -3,      /* call order violated (e.g. step before fields) */
  FPX_ERR_NOMEM = -4,
  FPX_ERR_UNSUPPORTED = -5
} fpx_status;

/* RNG modes.  TABLE_SEQ reproduces the reference bit-for-bit: the 1e6-entry
 * Gaussian table rannumb (com_mod.f90:744, filled as FLEXPART.f90:56-59) and a
 * start index per particle-step drawn from the shared sequential ran3 stream
 * in particle order (advance.f90:153, initialize.f90:68).  TABLE_COUNTER keeps
 * the table but draws the start index from a counter-based generator keyed on
 * (seed, particle id, step) -- order-independent, hence shardable.  PHILOX
 * replaces the table by Philox4x32-7 keyed on (seed, particle id, step, draw index / 4):
 * one call gives four clipped Box-Muller normals. */
typedef enum { FPX_RNG_TABLE_SEQ = 0, FPX_RNG_TABLE_COUNTER = 1, FPX_RNG_PHILOX = 2 } fpx_rng_mode;

typedef struct {
  int32_t struct_bytes;      /* = sizeof(fpx_config), ABI check                        */
  int32_t device;            /* HIP device ordinal                                     */
  int32_t compute_real_bytes;/* 8: all-fp64 arithmetic; 4: reference typing (f32, xy f64) */
  int32_t host_real_bytes;   /* default real kind of the host arrays (4 or 8)          */
  int64_t max_particles;     /* capacity (com_mod_allocate_part(nmpart), com_mod.f90:782) */
  /* grid: com_mod.f90:298-299 nx,ny,nz,nmixz,dx,dy,xlon0,ylat0; par_mod.f90:144 maxima */
  int32_t nx, ny, nz, nmixz;
  int32_t nxmax, nymax, nzmax;
  double dx, dy, xlon0, ylat0;
  /* com_mod.f90:551-560 */
  int32_t xglobal, nglobal, sglobal;
  double switchnorthg, switchsouthg;
  double northpolemap[9], southpolemap[9];
  /* run switches: com_mod.f90:56-77,112,144,188; derived in readcommand.f90:244-272,379-385 */
  int32_t ldirect, lsynctime, method, mintime, ifine, turbswitch, cblflag, mdomainfill, lsettling;
  double ctl;                /* com_mod ctl, i.e. already 1/CTL                         */
  double d_trop, d_strat, turbmesoscale;   /* par_mod.f90:79                           */
  /* species: com_mod.f90:170-189,589 (the release-point tables xmass, npart: fpx_set_release_points) */
  int32_t nspec, maxspec;    /* maxspec = species stride count of host xmass1           */
  int32_t drydep, drydepspec[FPX_MAXSPEC];
  double density[FPX_MAXSPEC], dquer[FPX_MAXSPEC], vsetaver[FPX_MAXSPEC], cunningham[FPX_MAXSPEC];
  double decay[FPX_MAXSPEC];
  int32_t mquasilag;         /* com_mod.f90:74,101: /= 0 skips the mass-fraction test, timemanager.f90:663 */
  int32_t lage_last;         /* lage(nageclass), com_mod.f90:129                        */
  /* RNG */
  int32_t rng_mode;          /* fpx_rng_mode                                            */
  uint64_t seed;             /* counter modes                                           */
  /* locality: re-sort particles by grid cell every `sort_interval` steps (0 = never) */
  int32_t sort_interval;
  /* par_mod nxmax of the HOST build, used for eps = nxmax/3.e5 (advance.f90:107); 0: use nxmax */
  int32_t par_nxmax;
  /* Several ranks (one handle per GPU, each holding a contiguous range of the run's particles, mpi_mod.f90:323):
   * number of this rank's first particle in the run's global numbering (0 on a single rank).  The counter RNG modes
   * key on the global number, and fpx_seed_particles generates that slice of the global synthetic cloud, so results
   * do not depend on how many ranks share the particles.  Particle indices at the boundary (first, count) stay local. */
  int64_t particle_base;
  /* Backward runs with receptor scavenging (COMMAND ind_receptor = 4 / 3; com_mod.f90:591 DRYBKDEP, WETBKDEP;
   * readcommand.f90:320-340): the particle array xscav_frac1 exists, the first step of a particle fills it
   * (timemanager.f90:564-598) and fpx_conccalc weights every contribution with max(xscav_frac1, 0).  ldirect must be -1. */
  int32_t drybkdep, wetbkdep;
  /* The two compile-time logicals of com_mod.f90:777-778 as run-time switches (a host passes its own parameters):
   * turboff /= 0: no turbulence (advance.f90:464-470: wp = 0, delz = 0 in the vertical Langevin loop; :675-679: no
   * random displacement above the boundary layer; the shipped value is .false.);
   * interpolhmix /= 0: the mixing height of the cell is interpolated in time (advance.f90:240-249 / :254-262 for a
   * nest) instead of taken as the maximum over both times (shipped: .false.). */
  int32_t turboff, interpolhmix;
  /* Time-blended wind packs (u, v, w, rho, drhodz blended in time once per step, because the time
   * weights are the same for every particle).  The blended and the unblended gather round differently (1e-11), so the
   * choice must not depend on anything a rank sees alone: 0 = decide from global_particles (on from 3e7), 1 = on,
   * 2 = off.  Every rank of a run must pass the same values. */
  int32_t blend_mode;
  /* Time slices of the Langevin kernel (k_pbl_loop): a launch gives a particle at most this many passes of the loop
   * advance.f90:282-609, the particles that need more continue in the next launch, shared by all waves.
   * 0 = the engine's schedule, -1 = one launch without a budget, k > 0 = k passes per launch.  Results do not depend on it. */
  int32_t pbl_slice_passes;
  /* The run's particle count over ALL ranks (what maxpart / the planned releases amount to), 0 = max_particles of this
   * rank: the basis of decisions that every rank must take alike (blend_mode = 0). */
  int64_t global_particles;
  /* Diagnostics inside the particle loop that the engine does NOT compute (out of scope): the host passes
   * its COMMAND switches and fpx_create refuses (FPX_ERR_UNSUPPORTED) a run that would need them, instead of dropping their
   * output silently: ipout = 3 (partpos_average, timemanager.f90:617), iflux = 1 (calcfluxes, :623), linit_cond >= 1
   * (initial_cond_calc, :631,702).  Zero-initialised fields mean "not requested". */
  int32_t ipout, iflux, linit_cond;
  int32_t reserved[3];
} fpx_config;

/* One time slot of the met fields the path gathers from (com_mod.f90:355-371,
 * 423-451).  Each pointer addresses element (0,0,1,slot) of the host array,
 * e.g. c_loc(uu(0,0,1,memind(k))) -- strides nxmax, nymax, nzmax as declared
 * in fpx_config.  NULL is allowed for fields a configuration never reads
 * (uupol/vvpol without poles, tt without settling, vdep without DRYDEP). */
typedef struct {
  const void *uu, *vv, *ww, *uupol, *vvpol, *rho, *drhodz, *tt;   /* (nxmax,nymax,nzmax)  */
  const void *hmix, *ustar, *wstar, *oli, *tropopause;            /* (nxmax,nymax)        */
  const void *vdep;                                               /* (nxmax,nymax,maxspec)*/
} fpx_fields;

/* Particle SoA: com_mod.f90:678-695.  reals other than xtra1/ytra1 have
 * host_real_bytes each.  NULL members are skipped on upload (defaults: 0, cbt 1,
 * npoint/nclass 1) and on download. */
typedef struct {
  double *xtra1, *ytra1;
  void *ztra1, *uap, *ucp, *uzp, *us, *vs, *ws;
  int32_t *itra1, *itramem, *idt, *npoint, *nclass;
  int16_t *cbt;
  void *xmass1;            /* (count or ld, nspec) species-major, leading dim xmass1_ld */
  int64_t xmass1_ld;
  int32_t *itrasplit;      /* com_mod.f90:683: next time the particle is split (NULL: never, i.e. ldirect*999999999, the value of a fresh engine) */
  void *xscav_frac1;       /* com_mod.f90:683 xscav_frac1(maxpart,maxspec), laid out like xmass1 (leading dim xmass1_ld); only with
                              drybkdep / wetbkdep.  NULL on upload: -1 (not yet scavenged, releaseparticles.f90:167-171) */
} fpx_particles;

typedef struct {
  int64_t n_due;           /* particles with itra1 == itime (advance calls made)     */
  int64_t n_initialized;   /* of those, newly released (initialize calls)            */
  int64_t n_left_domain;   /* nstop == 3, advance.f90:806                            */
  int64_t n_min_mass;      /* timemanager.f90:681-686                                */
  int64_t n_max_age;       /* timemanager.f90:701-707                                */
  int64_t nan_count;       /* CBL re-initialisations, advance.f90:421                */
  int64_t nan_count2;      /* advance.f90:439                                        */
  int64_t n_bad_position;  /* non-finite position caught before a gather (terminated) */
  double kernel_ms;        /* device time of the step's kernels (HIP events)         */
} fpx_step_stats;

/* ---- life cycle ----------------------------------------------------------- */
int fpx_create(fpx_handle *out, const fpx_config *cfg);
int fpx_destroy(fpx_handle h);
const char *fpx_last_error(void);
int fpx_abi_version(void);

/* northpolemap/southpolemap (com_mod.f90:560) for a global grid with latitude spacing dy,
 * computed as the reference's grid check does (gridcheck_ecmwf.f90:341-366 via cmapf_mod
 * stlmbr/stcm2p) in the host's real kind.  Pure host helper for callers that do not have
 * the reference's own records; a Fortran host passes its com_mod arrays instead. */
int fpx_polar_maps(int32_t host_real_bytes, double dy, double north[9], double south[9]);

/* height(1:nz) of com_mod.f90:299 (host_real_bytes each) */
int fpx_set_height(fpx_handle h, const void *height, int32_t n);

/* ---- met fields ----------------------------------------------------------- */
/* Upload one time slot (slot = 1 or 2, the value found in memind()) and repack
 * it into the device layout.  Replaces the implicit use of com_mod's field
 * arrays after getfields() (getfields.f90:81-225). */
int fpx_upload_fields(fpx_handle h, int32_t slot, const fpx_fields *f);
/* ---- verttransform_ecmwf on the device (SURVEY section 8 f, item 1) -------------------------
 * Replaces `call verttransform_ecmwf(memind(2),uuh,vvh,wwh,pvh)` (getfields.f90:129,164,180; the
 * routine: verttransform_ecmwf.f90:55-590) followed by fpx_upload_fields: the model-level arrays
 * readwind_ecmwf has filled go to the device, the eta -> z transform (uvzlev/wzlev, pinmconv,
 * vertical interpolation of u,v,T,q,pv,rho, w conversion, drhodz, eta-slope correction, polar
 * stereographic winds) runs there and its result is repacked straight into the gather layout.
 * The cloud diagnostics of verttransform_ecmwf.f90:604-880 and `prs` are not computed (not read
 * by the particle path; a host that needs them keeps its own loop over `out->tt, qv, rho`).
 * All pointers address host arrays in the reference's own shapes. */
typedef struct {
  const void *uuh, *vvh, *pvh;   /* (0:nxmax-1,0:nymax-1,nuvzmax), verttransform_ecmwf.f90:62  */
  const void *wwh;               /* (0:nxmax-1,0:nymax-1,nwzmax),  :63                          */
  const void *tth, *qvh;         /* c_loc(tth(0,0,1,n)), com_mod.f90:372-373                    */
  const void *ps, *tt2, *td2;    /* c_loc(ps(0,0,1,n)) ..., com_mod.f90:410-417                 */
  const void *akz, *bkz;         /* (nuvz), com_mod.f90:329                                     */
  const void *aknew, *bknew;     /* (nz),   com_mod.f90:330                                     */
  int32_t nuvz, nwz;             /* must equal nz (gridcheck_ecmwf.f90 sets nz = nuvz)          */
  int32_t init;                  /* 1: derive height(nz) and nmixz from this input as the first
                                    call of the reference does (:118-196); 0: use fpx_set_height's.
                                    Without fpx_set_height the first call derives them anyway.  */
  int32_t pin_host;              /* 1: the input arrays keep their addresses until fpx_destroy (static com_mod
                                    arrays): register them once for DMA (hipHostRegister) -- the 433 MB
                                    host-to-device copy then runs at PCIe speed instead of the pageable path */
  double nest_dy, nest_ylat0;    /* fpx_verttransform_nest only: dyn(l), ylat0n(l) of the nest (com_mod.f90:477-479),
                                    used in cosf (verttransform_nests.f90:346)                       */
} fpx_model_levels;
/* Optional copies back to the host (NULL members are skipped): the z-level arrays in the host's
 * shapes (0:nxmax-1,0:nymax-1,nzmax) for slot n, e.g. c_loc(tt(0,0,1,n)); height(nz); nmixz. */
typedef struct {
  void *uu, *vv, *ww, *tt, *qv, *pv, *rho, *drhodz, *uupol, *vvpol;
  void *height;
  int32_t *nmixz;
} fpx_fields_out;
/* sfc: the 2-D members of fpx_fields (hmix, ustar, wstar, oli, tropopause, vdep) when calcpar runs on the host; its
 * 3-D members are ignored.  sfc = NULL: fpx_calcpar(h, slot, ..) follows and computes them on the device. */
int fpx_verttransform_ecmwf(fpx_handle h, int32_t slot, const fpx_model_levels *m, const fpx_fields *sfc, const fpx_fields_out *out);
/* The same for nested grid `nest` (1-based, after fpx_nests_init): replaces `call verttransform_nests(memind(k),
 * uuhn,vvhn,wwhn,pvhn)` (getfields.f90:133,168,184; verttransform_nests.f90:55-420 without its cloud diagnostics) and the
 * fpx_upload_nest_fields of that slot.  Pointers address the nest's arrays of that nest, e.g. c_loc(uuhn(0,0,1,l)),
 * c_loc(tthn(0,0,1,n,l)), strides nxmaxn, nymaxn; the z levels are the mother grid's. */
int fpx_verttransform_nest(fpx_handle h, int32_t nest, int32_t slot, const fpx_model_levels *m, const fpx_fields *sfc, const fpx_fields_out *out);
/* device time of the transform kernels of the last call, milliseconds */
int fpx_verttransform_time(fpx_handle h, double *ms);
/* ---- calcpar on the device (SURVEY section 8 f, item 1, second half) -------------------------------------
 * Replaces `call calcpar(n,uuh,vvh,pvh)` (getfields.f90:128,163,179; the routine: calcpar.f90:76-265, ECMWF branch) for the
 * fields the particle path reads: ustar (scalev.f90), oli (obukhov.f90), hmix and wstar (richardson.f90 with qvsat.f90,
 * hmixmin/hmixmax, the subgrid-orography excess with lsubgrid = 1) and the thermal tropopause (:199-265).  It runs on the
 * model-level arrays fpx_verttransform_ecmwf(h, slot, m, sfc, ..) uploaded for this slot -- call that first, with
 * sfc = NULL (then the 2-D fields come from here and nothing but ps, tt2, td2, surfstr, sshf crosses PCIe) -- and
 * writes the gather packs of the slot directly.  Not computed: vdep (getvdep, calcpar.f90:174-193: the land-use tables
 * stay with the host; with DRYDEP pass the host's vdep of this slot) and the potential vorticity (calcpv, :270).
 * The reference calls calcpar before verttransform; the two are independent of each other's results. */
typedef struct {
  const void *surfstr, *sshf;    /* c_loc(surfstr(0,0,1,n)), c_loc(sshf(0,0,1,n)), com_mod.f90:420-422          */
  const void *akm, *bkm;         /* (nwz) com_mod.f90:328                                                       */
  const void *excessoro;         /* (0:nxmax-1,0:nymax-1) com_mod.f90:343, read with lsubgrid = 1 (else NULL)   */
  const void *vdep;              /* c_loc(vdep(0,0,1,n)) from the host's getvdep, required with DRYDEP          */
  int32_t lsubgrid;              /* com_mod.f90:117                                                             */
  int32_t reserved[3];
} fpx_calcpar_in;
/* optional copies back into the host's arrays of slot n (NULL members are skipped), e.g. c_loc(hmix(0,0,1,n)) */
typedef struct { void *ustar, *wstar, *oli, *hmix, *tropopause; } fpx_calcpar_out;
int fpx_calcpar(fpx_handle h, int32_t slot, const fpx_calcpar_in *c, const fpx_calcpar_out *out);
/* device time of the kernel of the last fpx_calcpar call, milliseconds */
int fpx_calcpar_time(fpx_handle h, double *ms);
/* ---- partoutput: the binary particle dump (SURVEY section 8 f, item 4) ----------------------
 * Replaces `call partoutput(itime)` (timemanager.f90:454; the routine: partoutput.f90:63-190):
 * for every particle with itra1 == itime the device interpolates oro, pv, qv, tt, rho, hmix and
 * tropopause to the particle position exactly as the routine does and builds the records of the
 * file partposit_<date> / partposit_end (Fortran sequential unformatted, 4-byte record markers) in
 * particle-number order; the host streams the bytes to `path`.  The file is byte-identical to the
 * reference's, in the host's real kind.  No particle array travels to the host.
 * The fields the particle path itself does not use are kept on the device in the host's layout:
 * oro (0:nxmax-1,0:nymax-1) com_mod.f90:342 and pv, qv, tt (0:nxmax-1,0:nymax-1,nzmax) of a slot,
 * com_mod.f90:360-369.  fpx_verttransform_ecmwf retains pv, qv, tt of its slot by itself; slot 0
 * uploads oro only.  NULL members are skipped. */
typedef struct {
  const void *oro;
  const void *pv, *qv, *tt;
} fpx_diag_fields;
int fpx_upload_diag_fields(fpx_handle h, int32_t slot, const fpx_diag_fields *f);
/* the same for nested wind field `nest` (after fpx_nests_init; strides nxmaxn, nymaxn): oron(0:nxmaxn-1,0:nymaxn-1,nest) with any
 * slot, ttn(:,:,:,2,nest) with slot 2 -- what releaseparticles.f90:216-273 reads for a particle released inside a nest (its rhon
 * comes from fpx_upload_nest_fields / fpx_verttransform_nest); pv, qv are ignored */
int fpx_upload_diag_nest_fields(fpx_handle h, int32_t nest, int32_t slot, const fpx_diag_fields *f);
/* nparticles (may be NULL): number of particle records written */
int fpx_partoutput(fpx_handle h, int32_t itime, const char *path, int64_t *nparticles);
/* device time (selection, scan, record kernel) of the last fpx_partoutput call, milliseconds */
int fpx_partoutput_time(fpx_handle h, double *ms);
/* ---- readpartpositions: warm start from a dump (SURVEY section 8 f, item 4) -------------------
 * Replaces the part of `call readpartpositions` (FLEXPART.f90; the routine: readpartpositions.f90:115-148)
 * that follows the `header` checks: the file partposit_end (one dump as partoutput / fpx_partoutput write
 * it) is streamed to the device and parsed there into the particle arrays: xtra1 = (xlon-xlon0)/dx,
 * ytra1, ztra1, npoint, xmass1, itramem re-based on this run's start, idt = mintime, itra1 = 0,
 * nclass (ran1 stream, seed -8, when nclassunc > 1), turbulent state zero (the first step calls
 * initialize() for every particle, timemanager.f90:553), itrasplit = ldirect*itsplit (:117).  The host keeps
 * reading `header` (:59-113) and passes what it found. */
typedef struct {
  double jul_header;   /* juldate(ibdatein,ibtimein) of the header file, readpartpositions.f90:133 */
  double bdate;        /* com_mod bdate: start of this run                                        */
  int32_t mintime;     /* com_mod.f90:112                                                          */
  int32_t nclassunc;   /* par_mod.f90:188                                                          */
  int32_t itrasplit;   /* ldirect*itsplit: the value readpartpositions.f90:117 gives itrasplit(i)   */
  int32_t reserved;
} fpx_restart;
int fpx_readpartpositions(fpx_handle h, const char *path, const fpx_restart *r,
                          int64_t *numpart, int32_t *numparticlecount, int32_t *itimein);
/* ---- convective mixing of the particles (SURVEY section 8 f, item 3) ------------------------------
 * Replaces `call convmix(itime,metdata_format)` (timemanager.f90:258-262 forward, :183-187 backward; the routine:
 * convmix.f90:61-196, ECMWF input, mother grid) with everything below it: calcmatrix.f90:56-137 per grid column that
 * holds particles, Emanuel's scheme CONVECT / TLIFT (convect43c.f90) and redist.f90:49-236 per particle.
 * fpx_conv_init: the level structure (com_mod nuvz, conv_mod nconvlev, akz, bkz, akm, bkm of gridcheck_ecmwf.f90);
 *   allocates the cloud-base mass flux cbaseflux(0:nxmax-1,0:nymax-1) (conv_mod.f90), zero as at start-up.
 * fpx_upload_conv_fields: ps, tt2, td2 (2-D) and tth, qvh (nuvzmax levels) of one wind-field slot, host layout
 *   (nxmax, nymax strides), as readwind_ecmwf leaves them -- the same arrays fpx_verttransform_ecmwf takes.
 * fpx_convmix: moves the particles that are due at itime; nmoved (may be NULL): particles whose height was set.
 *   Random numbers: the serial ran3 stream shared with advance / initialize (redist.f90:69,130: its own seed -88 re-seeds
 *   the shared generator at the first call) replayed by the host in the order of the reference's sort2 in the parity
 *   mode FPX_RNG_TABLE_SEQ; one counter-generator draw per particle and step otherwise.
 * fpx_get_cbaseflux / fpx_set_cbaseflux: the mass-flux field, compact [ny][nx] in the host's real kind (restart files).
 * Nested wind fields (convmix.f90:100-119,198-250; after fpx_nests_init): a particle inside a nest takes the nest's columns,
 *   soundings and mass-flux field cbasefluxn(:,:,l); fpx_upload_conv_nest_fields(nest, slot, ...) with the strides nxmaxn,
 *   nymaxn of fpx_nests_init is then required for every nest, fpx_get/set_cbaseflux_nest move [nyn][nxn].
 * Not covered: the flux diagnostics of calcfluxes (iflux = 1). */
typedef struct {
  int32_t struct_bytes;
  int32_t nuvz;              /* com_mod nuvz                                              */
  int32_t nconvlev;          /* conv_mod nconvlev (gridcheck_ecmwf.f90:560-565)            */
  int32_t reserved;
  const void *akz, *bkz;     /* [nuvz] host real: full levels (akz(1) = 0, bkz(1) = 1)     */
  const void *akm, *bkm;     /* [nuvz] host real: half levels                              */
} fpx_conv_config;
typedef struct {
  const void *ps, *tt2, *td2;   /* (0:nxmax-1,0:nymax-1)            */
  const void *tth, *qvh;        /* (0:nxmax-1,0:nymax-1,nuvzmax)    */
  int32_t nuvzmax;              /* allocated levels of tth, qvh     */
  int32_t reserved;
} fpx_conv_fields;
int fpx_conv_init(fpx_handle h, const fpx_conv_config *c);
int fpx_upload_conv_fields(fpx_handle h, int32_t slot, const fpx_conv_fields *f);
int fpx_convmix(fpx_handle h, int32_t itime, int64_t *nmoved);
int fpx_convmix_time(fpx_handle h, double *ms);
int fpx_upload_conv_nest_fields(fpx_handle h, int32_t nest, int32_t slot, const fpx_conv_fields *f);
int fpx_get_cbaseflux_nest(fpx_handle h, int32_t nest, void *cbasefluxn);
int fpx_set_cbaseflux_nest(fpx_handle h, int32_t nest, const void *cbasefluxn);
int fpx_get_cbaseflux(fpx_handle h, void *cbaseflux);
int fpx_set_cbaseflux(fpx_handle h, const void *cbaseflux);
/* ---- lossless checkpoint (SURVEY section 8 f, item 4, last clause) -----------------------------
 * The reference's restart is lossy: partoutput.f90:63-190 writes position, mass and age in the dump's
 * real kind and omits uap..uzp, us..ws, cbt, idt, itramem, nclass; readpartpositions.f90:118-148 sets
 * them anew, so a warm-started run is another realisation of the run.  This pair has no reference
 * counterpart: fpx_checkpoint_write stores every array of the particle loop (com_mod.f90:678-695) in the
 * compute precision and in particle-number order, the step counter the counter RNG is keyed on, the state
 * of the serial ran3 / ran1 streams of the parity mode and the accumulating grids (gridunc, drygridunc,
 * wetgridunc, the nested ones, creceptor) and the convection scheme's cbaseflux; fpx_checkpoint_read restores them into an engine created with
 * the same fpx_config (+ fpx_outgrid_init ... if the run samples), also after a locality sort.  A run continued
 * from the file is bit-identical to the uninterrupted one.  itime and numparticlecount are the host's values
 * and come back unchanged.  The met fields are not in the file (the host uploads them as at start-up). */
int fpx_checkpoint_write(fpx_handle h, const char *path, int32_t itime, int32_t numparticlecount);
int fpx_checkpoint_read(fpx_handle h, const char *path, int32_t *itime, int64_t *numpart, int32_t *numparticlecount);
/* ---- concoutput: the sparse concentration files (SURVEY section 8 f, item 4) -----------------
 * Replaces the part of `call concoutput(itime,outnum,...)` (timemanager.f90:384; the routine:
 * concoutput.f90:226-228,296-447) that writes grid_conc_<date><time>_<species> for a forward run with
 * iout = 1: per species, point-release class and age class the class mean of wetgridunc, drygridunc and
 * gridunc and their run-length compressed dumps, computed on the device from the grids the sampling kernels
 * filled; only the compressed indices and values travel to the host, which writes the records.
 * Byte-identical to the reference's file for the same grids.  Hosts with a 4-byte default real only (the
 * reference's concoutput.f90 does not compile with -fdefault-real-8).  Not written: dates,
 * factor_drygrid, the receptor files (host side, from fpx_get_receptors).  For the sums over all ranks call
 * fpx_get_grids(h, NULL, NULL, 1, 0) / fpx_get_wetgrid(h, NULL, 1) (/ fpx_get_grids_nest) first and set
 * fpx_concout.reduced = 1: the files are then written from the receive buffers of those reductions
 * (concoutput_mpi.f90:279,298 reads gridunc0, drygridunc0, wetgridunc0), on the rank(s) that call this.
 * prefix: the file name without the species number, e.g. "<path>grid_conc_20200101010000_".
 * clear = 1 zeroes gridunc afterwards as the routine does (:714). */
typedef struct {
  const void *area;      /* area(0:numxgrid-1,0:numygrid-1), outg_mod (outgrid_init.f90:59-84)          */
  const void *volume;    /* volume(0:numxgrid-1,0:numygrid-1,numzgrid)                                  */
  double outnum;         /* sum of the sampling weights of the averaging interval (timemanager.f90:363)  */
  int32_t wetdep, drydep;
  int32_t nest;          /* 1: the nested output grid (`call concoutput_nest`, timemanager.f90:418; concoutput_nest.f90): area and
                            volume are arean, volumen; the prefix is "<path>grid_conc_nest_<date><time>_"                          */
  int32_t iout;          /* 0 or 1: grid_conc_* only; 2: grid_pptv_* only; 3: both (com_mod iout, concoutput.f90:265,482)     */
  /* mixing-ratio files (iout 2, 3): "<path>grid_pptv_[nest_]<date><time>_", outheight(numzgrid),
   * outlon0, outlat0 (com_mod.f90:583-584; outlon0n, outlat0n with nest = 1) and weightmolar(1:nspec) (com_mod.f90:177); the air density comes from the
   * met fields of slot memind(2) on the device (densityoutgrid, concoutput.f90:176-205) */
  const char *prefix_pptv;
  const void *outheight;
  double outlon0, outlat0;
  double weightmolar[FPX_MAXSPEC];
  int32_t reduced;       /* 1: write the sums over all ranks left by the preceding allreduce calls (see above); 0: this rank's own grids */
  int32_t reserved;
} fpx_concout;
int fpx_concoutput(fpx_handle h, int32_t itime, const fpx_concout *c, const char *prefix, int32_t clear);
/* memtime(1:2), memind(1:2) of com_mod.f90:286; lwindinterv = |memtime(2)-memtime(1)|. */
int fpx_set_windtime(fpx_handle h, const int32_t memtime[2], const int32_t memind[2]);

/* ---- nested grids (com_mod.f90:464-541; geometry from gridcheck_nests.f90:362-378) -------- */
typedef struct {
  int32_t struct_bytes;
  int32_t numbnests;
  int32_t nxmaxn, nymaxn;                 /* allocated extents of the host nest arrays (par_mod) */
  int32_t nxn[FPX_MAXNESTS], nyn[FPX_MAXNESTS];
  double xln[FPX_MAXNESTS], yln[FPX_MAXNESTS], xrn[FPX_MAXNESTS], yrn[FPX_MAXNESTS];
  double xresoln[FPX_MAXNESTS], yresoln[FPX_MAXNESTS];   /* entries 1..numbnests of the host arrays */
} fpx_nests;
int fpx_nests_init(fpx_handle h, const fpx_nests *n);
/* One time slot of one nest (nest = 1..numbnests): pointers to element (0,0,1,slot,nest) of
 * uun, vvn, wwn, rhon, drhodzn (com_mod.f90:501-502), hmixn, ustarn, wstarn, olin, tropopausen
 * (:523-527), vdepn (:529); strides nxmaxn, nymaxn, nzmax.  uupol/vvpol/tt are ignored. */
int fpx_upload_nest_fields(fpx_handle h, int32_t nest, int32_t slot, const fpx_fields *f);

/* ---- RNG ------------------------------------------------------------------ */
/* Build rannumb exactly as FLEXPART.f90:47,56-59 does (seed -320, gasdev1 over
 * the integer ran3 generator) in host_real_bytes precision, keep the ran3
 * stream state for the per-step start indices, and upload the table. */
int fpx_rng_fill_table(fpx_handle h);
/* or install a table made by the caller (rannumb(1:maxrand)) */
int fpx_rng_set_table(fpx_handle h, const void *rannumb, int32_t maxrand);
/* copy of the table (tests) */
int fpx_rng_get_table(fpx_handle h, void *rannumb, int32_t maxrand);

/* ---- particles ------------------------------------------------------------ */
/* Copy particles [first, first+count) from the host SoA (after releaseparticles,
 * particle splitting or a warm start) / back to it. Indices are the reference's
 * particle numbers j-1; an internal locality sort never changes them.
 * A particle is initialised (initialize.f90) in the step whose time equals its itramem
 * (timemanager.f90:553) -- also one that the host uploads ahead of its birth (itra1 =
 * itramem = a later time): it is not due before, and the step at that time initialises it. */
int fpx_upload_particles(fpx_handle h, int64_t first, int64_t count, const fpx_particles *p);
int fpx_download_particles(fpx_handle h, int64_t first, int64_t count, const fpx_particles *p);
int fpx_set_numpart(fpx_handle h, int64_t numpart);   /* com_mod numpart */
/* The release-point tables of point_mod (point_mod.f90:10,20; allocated and filled by readreleases.f90:239-243,419-421,450-455):
 * xmass(numpoint, maxspec) in the host's real kind (column-major: species stride = numpoint) and
 * npart(numpoint).  The particle loop indexes both by the particle's release point npoint(j):
 * the mass-fraction termination real(npart(npoint(j)))*xmass1(j,ks)/xmass(npoint(j),ks) < minmass
 * (timemanager.f90:663-666,681-686) and the species whose settling velocity a particle takes,
 * the first with xmass(nrelpoint,nsp) > eps3 (advance.f90:518-531,686-699,893-906).  Required before
 * fpx_step unless mdomainfill /= 0 or mquasilag /= 0 (then only the settling pick reads it; without
 * the tables every particle takes species 1).  A particle whose npoint lies outside 1..numpoint
 * (the reference would read outside the arrays) is treated as a particle of the nearest point. */
int fpx_set_release_points(fpx_handle h, int32_t numpoint, const void *xmass, const int32_t *npart);
/* point_mod zpoint1(numpoint), zpoint2(numpoint) in the host's real kind: the height range of every release point.  Needed by
 * WETBKDEP only (xscav_frac1 = wetscav * (zpoint2 - zpoint1) * grfraction(1), timemanager.f90:590-591); fpx_release_init
 * sets it from its own tables. */
int fpx_set_release_heights(fpx_handle h, int32_t numpoint, const void *zpoint1, const void *zpoint2);

/* ---- releaseparticles and particle splitting on the device (SURVEY section 8 f, item 2) --------------------
 * fpx_releaseparticles replaces `call releaseparticles(itime)` (timemanager.f90:246; the routine:
 * releaseparticles.f90:63-375): for every release point that is active at itime the number of particles of this
 * step (local-time emission factors, rfraction / xmasssave bookkeeping: :63-128, on the host in the host's real kind),
 * then, on the device, per new particle: the k-th particle released takes the k-th storage space with
 * itra1 /= itime in particle-number order (:133-137,363-367), a random position in the release volume (:139-146,183),
 * its mass per species (:156-157), nclass, npoint, idt, itra1, itramem, itrasplit (:168-181), topography under the
 * particle and the conversions for kindz = 2 and 3 (:206-280, rho and tt of the literal time slot 2) and the density
 * factor for ind_rel = 1, 3, 4 (:300-341).  Arithmetic in the host's real kind without contraction: with the serial
 * random stream (rng mode FPX_RNG_TABLE_SEQ: ran1, seed -7, four draws per particle replayed on the host) every
 * array equals the reference's bit for bit; in the counter modes the four uniforms of a particle come from
 * Philox keyed on (seed, numparticlecount of the particle).  Particles never visit the host.
 * Needs: fpx_set_release_points (xmass, npart), oro (fpx_upload_diag_fields slot 0); for kindz = 3 also tt of
 * slot 2 (fpx_upload_diag_fields slot 2 or fpx_verttransform_ecmwf); rho comes from the met fields of slot 2.
 * Nested met grids (releaseparticles.f90:196-226): a particle released inside a nest takes the nest's orography, density and
 * temperature (fpx_upload_diag_nest_fields + the nest's field pack).
 * With fpx_config.drybkdep / wetbkdep a new particle gets xscav_frac1 = -1 ("not yet through the receptor block", :167-171). */
typedef struct {
  int32_t struct_bytes;
  int32_t numpoint;
  const int32_t *ireleasestart, *ireleaseend;            /* (numpoint) point_mod.f90:8-9                      */
  const int16_t *kindz;                                  /* (numpoint) integer*2, point_mod.f90:11            */
  const void *xpoint1, *xpoint2, *ypoint1, *ypoint2;     /* (numpoint) grid coordinates, point_mod.f90:13-16  */
  const void *zpoint1, *zpoint2;                         /* (numpoint) point_mod.f90:17-18                    */
  const void *point_hour, *area_hour;                    /* (maxspec,24) com_mod.f90:184                      */
  const void *point_dow, *area_dow;                      /* (maxspec,7)  com_mod.f90:185                      */
  double bdate;                                          /* com_mod.f90:45 (juldate of the run's start)       */
  int32_t itsplit, ind_rel, nclassunc;                   /* com_mod.f90:112,75; par_mod.f90:188               */
  int32_t reserved[5];
} fpx_release;
/* The tables are copied; fpx_set_release_points must have been called with the same numpoint. */
int fpx_release_init(fpx_handle h, const fpx_release *r);
/* numpart, numparticlecount (com_mod.f90:676,678) and xmasssave(numpoint) (xmass_mod) are the host's own variables,
 * read and updated like the routine does; rho_rel(numpoint) (point_mod.f90:21) is written for the points that
 * released with ind_rel = 1, 3, 4 (may be NULL).  nreleased (may be NULL): particles released by this call.
 * More particles than free storage spaces: FPX_ERR_NOMEM and nothing is released (the reference stops, :369-378). */
int fpx_releaseparticles(fpx_handle h, int32_t itime, int64_t *numpart, int32_t *numparticlecount, void *xmasssave,
                         void *rho_rel, int64_t *nreleased);
/* The splitting block of the time manager, timemanager.f90:473-504: every particle j <= numpart with
 * ldirect*itime >= ldirect*itrasplit(j) is copied to the next storage space behind numpart (in the order of j, while
 * there is room), both halves carry half the mass and the doubled splitting interval.  numpart in/out. */
int fpx_split_particles(fpx_handle h, int32_t itime, int64_t *numpart);

/* ---- keeping the ranks' particle counts level (mpi_mod.f90:566-856) ---------------------------------------
 * mpif_calculate_part_redist (:566-658): every rank knows all counts (the host's MPI_Allgather of numpart, :603); the
 * counts are sorted, the rank with the most particles is paired with the one with the fewest, the second with the
 * second-fewest ...; a pair exchanges (difference)/2 particles when the larger count exceeds mp_min_redist = 100000 and
 * the difference more than mp_redist_fract = 0.2 of it; never with ipout = 3.  role: 0 nothing to do, 1 this rank sends,
 * 2 it receives num_trans particles from / to peer.  Pure host arithmetic: no handle, no device. */
int fpx_redist_plan(const int64_t *npart_per_process, int32_t nranks, int32_t rank, int32_t ipout, int32_t *role, int32_t *peer,
                    int64_t *num_trans);
/* mpif_redist_part (:661-856).  The host keeps the transport: ONE message of fpx_redist_bytes(h, num_trans) bytes instead
 * of the reference's 9 + nspec.  Sender (:700-746): the storage spaces numpart-num_trans+1 .. numpart are copied into buf
 * and terminated (itra1 = -999999999), numpart -= num_trans.  Receiver (:749-841): the received particles that are alive
 * at itime go, in order, into the storage spaces 1 .. numpart+num_trans whose itra1 /= itime; numpart = max(numpart, last
 * space used).  As in the reference only nclass, npoint, itra1, idt, itramem, itrasplit, xtra1, ytra1, ztra1 and xmass1
 * travel: the turbulent velocities, cbt (and xscav_frac1) of a storage space stay what its last owner left.
 * buf: host or device memory; layout (n = num_trans, R = the engine's compute real): xtra1 f64[n], ytra1 f64[n], ztra1 R[n],
 * xmass1 R[nspec][n], then int32[n] each: nclass, npoint, itra1, idt, itramem, itrasplit.  Both ranks must run the same
 * configuration.  numpart + num_trans beyond the capacity: FPX_ERR_NOMEM, nothing changed. */
uint64_t fpx_redist_bytes(fpx_handle h, int64_t num_trans);
int fpx_redist_pack(fpx_handle h, int32_t itime, int64_t num_trans, void *buf, uint64_t buf_bytes, int64_t *numpart);
int fpx_redist_unpack(fpx_handle h, int32_t itime, int64_t num_trans, const void *buf, uint64_t buf_bytes, int64_t *numpart);

/* ---- the hot path --------------------------------------------------------- */
/* One pass of the particle loop timemanager.f90:531-712 at time itime. */
int fpx_step(fpx_handle h, int32_t itime, fpx_step_stats *stats);
/* Same, asynchronous on the handle's stream and without statistics read-back
 * (benchmarks / graph capture).  fpx_sync() waits for it. */
int fpx_step_async(fpx_handle h, int32_t itime);
int fpx_sync(fpx_handle h);
/* counters accumulated over all (sync or async) steps since the last reset; kernel_ms is 0 */
int fpx_counters(fpx_handle h, fpx_step_stats *stats, int32_t reset);
/* cumulative device time (ms) and launch count of the advance kernel since the
 * last reset, measured with HIP events on the handle's stream */
int fpx_kernel_time(fpx_handle h, double *advance_ms, int64_t *launches, int32_t reset);
/* the same split per kernel: ms[0] = k_prep + the work-list sort, ms[1] = k_pbl_loop, ms[2] = k_pbl_finish,
 * ms[3] = k_prep alone (part of ms[0]) */
int fpx_kernel_times(fpx_handle h, double ms[4], int64_t *launches, int32_t reset);

/* Force a locality re-sort now (normally driven by cfg.sort_interval). */
int fpx_sort_particles(fpx_handle h);

/* Device-side synthetic cloud for benchmarks: n particles uniformly over the
 * grid from a SplitMix64 counter hash (same generator as
 * flexpart_amd/synthetic.py:make_particles), frac_pbl of them below the local
 * mixing height.  All due at itime0 and newly released. */
int fpx_seed_particles(fpx_handle h, int64_t n, uint64_t seed, double frac_pbl, double zmax,
                       double lat_margin_cells, int32_t itime0);

/* ---- concentration sampling and deposition grids ---------------- */
#define FPX_MAXAGECLASS 8
typedef struct {
  int32_t struct_bytes;
  int32_t numxgrid, numygrid, numzgrid;        /* com_mod.f90:583                          */
  double dxout, dyout, xoutshift, youtshift;   /* com_mod.f90:584; readoutgrid.f90:199-200 */
  int32_t maxpointspec_act, nclassunc, nageclass;   /* com_mod.f90:188, par_mod.f90:188    */
  int32_t lage[FPX_MAXAGECLASS];               /* com_mod.f90:129                          */
  int32_t ind_samp, ioutputforeachrelease;     /* com_mod.f90:74-75                        */
  int32_t lusekerneloutput;                    /* par_mod.f90:39                           */
  int32_t reserved[5];
} fpx_outgrid;

/* Allocate and zero the device copies of gridunc (unc_mod.f90:16, extents as allocated in
 * outgrid_init.f90:192: (0:numxgrid-1,0:numygrid-1,numzgrid,maxspec,maxpointspec_act,nclassunc,
 * nageclass)) and drygridunc (unc_mod.f90:25, no level dimension, real(dep_prec) = 4 bytes in
 * every build).  outheight(1:numzgrid): outg_mod.f90, host_real_bytes each. */
int fpx_outgrid_init(fpx_handle h, const fpx_outgrid *g, const void *outheight);
/* loutnext, loutstep (com_mod.f90:62): used for the decay correction of deposited mass,
 * timemanager.f90:513-517,654-655 */
int fpx_set_output_times(fpx_handle h, int32_t loutnext, int32_t loutstep);
/* conccalc(itime, weight), conccalc.f90:50-295: scatter-add of every particle with
 * itra1 == itime into the device gridunc.  Dry-deposited mass is accumulated into drygridunc by
 * fpx_step itself (drydepokernel.f90:41-116, called at timemanager.f90:690-696). */
int fpx_conccalc(fpx_handle h, int32_t itime, double weight);
/* Copy the grids to the host (gridunc: host_real_bytes per value, drygridunc: 4 bytes; either may be NULL).
 * allreduce != 0: the sum over all ranks of the communicator (the MPI_Reduce of mpi_mod.f90:2471-2492, as one
 * RCCL all-reduce over xGMI per grid, or through the host's own all-reduce: fpx_comm_init_host).  Like the reference
 * (gridunc0, drygridunc0, wetgridunc0: mpi_mod.f90:2484-2492, outgrid_init.f90:220-225) the engine reduces into
 * receive buffers of its own and copies out from those: every rank's partial sums stay untouched, because
 * drygridunc and wetgridunc accumulate over the whole run and are reduced again at the next output time.
 * clear != 0: zero the rank's gridunc afterwards, as concoutput does after writing (concoutput.f90:719-720: gridunc
 * and creceptor only -- the deposition grids are zeroed once, outgrid_init.f90:317-318, and never again). */
int fpx_get_grids(fpx_handle h, void *gridunc, void *drygridunc, int32_t allreduce, int32_t clear);
/* Communicator for the grid reduction, one of:
 * (a) RCCL: rank 0 obtains an id (128 bytes), the host distributes it (MPI_Bcast / torch.distributed / a file), every
 *     rank calls fpx_comm_init; the reduction then runs device to device on the handle's stream. */
int fpx_comm_unique_id(void *id, int32_t nbytes);
int fpx_comm_init(fpx_handle h, const void *id, int32_t nbytes, int32_t nranks, int32_t rank);
/* (b) the host's own all-reduce, e.g. MPI_Allreduce of the reference's MPI build (the transport mpi_mod.f90:2471-2492
 *     itself uses; works across nodes): fn(user, send, recv, count, dtype) sums `count` values of send over all ranks
 *     into recv (host buffers; dtype 0 = 4-byte real, 1 = 8-byte real) and returns 0 on success.  The engine stages
 *     the grids through pinned host memory.  Called once per grid from inside fpx_get_grids / fpx_get_wetgrid /
 *     fpx_get_grids_nest / fpx_get_receptors, in the same order on every rank. */
typedef int (*fpx_allreduce_fn)(void *user, const void *send, void *recv, int64_t count, int32_t dtype);
int fpx_comm_init_host(fpx_handle h, int32_t nranks, int32_t rank, fpx_allreduce_fn fn, void *user);
/* The particle count the reference's root process reduces at every output time (MPI_Reduce of numpart,
 * timemanager_mpi.f90:552-562): local[0] = live particles of this rank (itra1 /= -999999999), local[1] = its numpart;
 * total[] = the same summed over the ranks of the communicator when allreduce != 0 (RCCL: one ncclAllReduce of two
 * 64-bit integers on the handle's stream; host transport: the callback with two 8-byte reals), else a copy of local[].
 * Collective when allreduce != 0: every rank calls it, in the same order relative to the grid reductions. */
int fpx_count_particles(fpx_handle h, int64_t local[2], int64_t total[2], int32_t allreduce);

/* ---- wet deposition ------------------------------------------------- */
typedef struct {
  int32_t struct_bytes;
  int32_t wetdepspec[FPX_MAXSPEC];                       /* com_mod.f90:589 WETDEPSPEC   */
  double weta_gas[FPX_MAXSPEC], wetb_gas[FPX_MAXSPEC];   /* com_mod.f90:171              */
  double crain_aero[FPX_MAXSPEC], csnow_aero[FPX_MAXSPEC];   /* :172                     */
  double ccn_aero[FPX_MAXSPEC], in_aero[FPX_MAXSPEC];    /* :174                         */
  double henry[FPX_MAXSPEC];                             /* :175                         */
  int32_t readclouds;                                    /* :139                         */
  int32_t reserved[7];
} fpx_wet_config;
/* one time slot of the precipitation / cloud fields: lsprec, convprec, tcc (com_mod.f90:413-419),
 * ctwc (:384, only with readclouds), tt (:360) as host reals with strides nxmax,nymax(,nzmax);
 * clouds integer(1) (:379), cloudsh integer (:380) */
typedef struct {
  const void *lsprec, *convprec, *tcc, *ctwc, *tt;
  const int8_t *clouds;
  const int32_t *cloudsh;
} fpx_wet_fields;
int fpx_wet_init(fpx_handle h, const fpx_wet_config *w);     /* after fpx_outgrid_init */
int fpx_upload_wet_fields(fpx_handle h, int32_t slot, const fpx_wet_fields *f);
/* wetdepo(itime, ltsample, loutnext), wetdepo.f90:58-151 with get_wetscav.f90:78-314,
 * interpol_rain.f90:68-130 and wetdepokernel.f90:38-108: the separate particle loop the time
 * manager runs before getfields (timemanager.f90:164-169). */
int fpx_wetdepo(fpx_handle h, int32_t itime, int32_t ltsample, int32_t loutnext);
/* wetgridunc (unc_mod.f90:27, real(dep_prec) = 4 bytes), allreduce as fpx_get_grids (into wetgridunc0,
 * mpi_mod.f90:2486-2488).  Never cleared: it accumulates over the run. */
int fpx_get_wetgrid(fpx_handle h, void *wetgridunc, int32_t allreduce);

/* Precipitation / cloud / temperature fields of one nested grid and time slot: lsprecn, convprecn, tccn
 * (com_mod.f90:518-520), ctwcn (:503, only with readclouds_nest), ttn (:501), cloudsn integer(1) (:505), passed in the
 * members of fpx_wet_fields with strides nxmaxn, nymaxn(, nzmax); each pointer is element (0,0,1,slot,nest).  A particle
 * inside a nest is scavenged with the nest's fields (get_wetscav.f90:82-101,126-128,150-151,197-199), so with nests
 * configured fpx_wetdepo requires both slots of every nest. */
int fpx_upload_wet_nest_fields(fpx_handle h, int32_t nest, int32_t slot, const fpx_wet_fields *f, int32_t readclouds_nest);

/* ---- nested output grid and receptor points (conccalc.f90:301-441, :451-498;
 *      a22/a23: drydepokernel_nest.f90, wetdepokernel_nest.f90) ------------------------------------ */
typedef struct {
  int32_t struct_bytes;
  int32_t numxgridn, numygridn;                    /* com_mod.f90:585                                  */
  double dxoutn, dyoutn, xoutshiftn, youtshiftn;   /* com_mod.f90:586; readoutgrid_nest.f90            */
  int32_t reserved[4];
} fpx_outgrid_nest;
/* nested_output = 1: allocate and zero griduncn, drygriduncn, wetgriduncn (unc_mod.f90:24-28; levels, species,
 * point-spec, class and age extents as the mother grid, outgrid_init_nest.f90).  After fpx_outgrid_init.  From then
 * on fpx_conccalc, the step's dry-deposition epilogue and fpx_wetdepo also feed the nested grids
 * (conccalc.f90:300, timemanager.f90:694-696, wetdepo.f90:142). */
int fpx_outgrid_nest_init(fpx_handle h, const fpx_outgrid_nest *g);
/* griduncn (host real kind), drygriduncn / wetgriduncn (real(dep_prec) = 4 bytes); any pointer may be NULL.
 * allreduce: the sums over the ranks (mpi_mod.f90:2543-2569), through receive buffers as in fpx_get_grids;
 * clear: zero the rank's griduncn only (concoutput_nest.f90). */
int fpx_get_grids_nest(fpx_handle h, void *griduncn, void *drygriduncn, void *wetgriduncn, int32_t allreduce, int32_t clear);
/* Receptor points: xreceptor, yreceptor in grid coordinates and receptorarea (readreceptors.f90:88-92,
 * com_mod.f90:658-659), host real kind.  fpx_conccalc then also accumulates creceptor (conccalc.f90:451-498). */
int fpx_receptors_init(fpx_handle h, int32_t numreceptor, const void *xreceptor, const void *yreceptor, const void *receptorarea);
/* creceptor(ld, maxspec) of the host (com_mod.f90:660, ld = maxreceptor): rows 1..numreceptor of columns 1..nspec are
 * overwritten with the accumulated values (allreduce: summed over the ranks into creceptor0, mpi_mod.f90:2480-2484;
 * clear: zero the rank's creceptor, concoutput.f90:720). */
int fpx_get_receptors(fpx_handle h, void *creceptor, int32_t ld, int32_t allreduce, int32_t clear);

/* raw stream handle (hipStream_t) for callers that enqueue their own work */
void *fpx_stream(fpx_handle h);

/* Diagnostics: evaluates one of the engine's fp64 device math helpers (fpx_device.hpp: the 1-2 ulp
 * replacements of exp/log/sqrt/division used inside the Langevin loop) on n host values, on the
 * current device.  fn: 0 m_expp, 1 m_logp, 2 m_sqrtp, 3 m_rcp, 4 m_rsqrt, 5 x**0.333333333 and
 * 6 x**(-2*0.333333333) (m_cuberoot_parts), 7 m_erf_e(x, m_expp(-x*x)), 8 m_pow08,
 * 9 m_exp_tab, 10 m_log_abs, 11 m_rcbrt (the table-based helpers of the fine sub-step).  No reference counterpart; used by the parity tests to bound
 * the helpers against libm.  Returns 0 or a negative fpx_status. */
int fpx_math_probe(int32_t fn, const double *x, double *y, int64_t n);
/* Diagnostics of a library built with -DFPX_LANE_STATS (all zeros otherwise): for code region r of the Langevin kernel
 * out[2r] = how many times a wave executed it, out[2r+1] = the lanes that were active, summed.  Regions: 0 a pass,
 * 1 a fine sub-step, 2 its CBL branch, 3 the Gaussian branch under cblflag, 4 the exponential-form branch, 5/6/7 hanna_short
 * neutral / unstable / stable, 8 lane refill, 9 hand-over of a finished particle, 10 iterations of the kernel's outer loop.
 * n <= 32 values are written; reset != 0 zeroes the counters. */
int fpx_lane_stats(fpx_handle h, uint64_t *out, int32_t n, int32_t reset);

/* Tuning and diagnostic knobs of one handle (none of them changes a result; measurements and A/B runs only -- the library
 * reads no environment variable).  Names: "verbose" (0|1: the engine reports its launch geometry on stderr),
 * "pbl_blocks_per_cu" (1..: fewer resident blocks of the Langevin kernel), "pbl_slices" (comma-separated pass budgets of the
 * successive launches of the Langevin kernel, 0 = no budget, e.g. "48,96,0"; overrides fpx_config.pbl_slice_passes),
 * "prep_lds_pad" (bytes of unused dynamic LDS of k_prep: lowers its occupancy), "prep_init_always" (0|1: every step runs the
 * instance of k_prep that can initialize() new particles, as in a run with a continuous release), "pbl_drain_lanes", "pbl_cost_buckets", "permute" ("soa"|"record"|"auto": the
 * permutation kernel of the locality sort), "vt_unfused" (0|1: the level-parallel chain of fpx_verttransform_ecmwf instead
 * of the fused tile kernels), "conv_scratch_mb", "conv_one_lane", "conv_no_walk", "conv_rows_plain" (fpx_convmix variants).
 * Unknown names and malformed values return FPX_ERR_ARG. */
int fpx_set_option(fpx_handle h, const char *name, const char *value);
/* What the engine decided or did, by name: "time_blended_packs" (1 when the steps of this handle blend the wind packs in
 * time: fpx_config.blend_mode / global_particles), "blended_steps" (steps that did so far), "pbl_launches_per_step"
 * (launches of the Langevin kernel per step = time slices + 1), "pbl_grid" (its persistent grid, blocks; 0 before the
 * first step), "pbl_blocks_per_cu" (resident blocks of four waves per CU the grid was sized for).  Unknown names return
 * FPX_ERR_ARG. */
int fpx_get_info(fpx_handle h, const char *name, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* FLEXPART_AMD_H */
