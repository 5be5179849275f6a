! flexgpu_mod -- ISO_C_BINDING shim between the FLEXPART Fortran host and the
! MI355X particle-advection engine (include/flexpart_amd.h).
!
! This is the file a maintainer adds to the reference's src/ (and to MODOBJS in
! its makefile).  It marshals the module variables the replaced block reads --
! the particle loop of timemanager.f90:531-712 -- into the explicit arguments of
! the C ABI.  The host arrays are passed by address (c_loc): no copies are made
! on the Fortran side, and the strides nxmax/nymax/nzmax of the static com_mod
! arrays are honoured by the engine.
!
! Typical use inside timemanager (see INTEGRATION.md):
!     call flexgpu_init(ierr)                       ! once, after gridcheck/readcommand
!     call flexgpu_upload_fields(memind(1)); call flexgpu_upload_fields(memind(2))
!     call flexgpu_set_windtime()                   ! after every getfields()
!     call flexgpu_upload_particles(1, numpart)     ! after releaseparticles / splitting
!     call flexgpu_step(itime, stats)               ! replaces "do j=1,numpart ... end do"
!     call flexgpu_download_particles(1, numpart)   ! before conccalc/partoutput need them
module flexgpu_mod
  use iso_c_binding
  use par_mod
  use com_mod
  use point_mod, only: xmass, npart, ireleasestart, ireleaseend, kindz, xpoint1, xpoint2, ypoint1, ypoint2, zpoint1, zpoint2, rho_rel
  use xmass_mod, only: xmasssave
  use outg_mod, only: outheight, area, volume
  use unc_mod, only: gridunc, drygridunc, wetgridunc, griduncn, drygriduncn, wetgriduncn
  use conv_mod, only: nconvlev, cbaseflux, cbasefluxn
  implicit none
  private
  public :: fpx_step_stats, flexgpu_init, flexgpu_finalize, flexgpu_upload_fields, &
            flexgpu_set_windtime, flexgpu_upload_particles, flexgpu_download_particles, &
            flexgpu_step, flexgpu_use_table_rng, flexgpu_handle, flexgpu_last_error, &
            flexgpu_outgrid_init, flexgpu_conccalc, flexgpu_get_grids, &
            flexgpu_wet_init, flexgpu_upload_wet_fields, flexgpu_wetdepo, flexgpu_verttransform, &
            flexgpu_upload_diag_fields, flexgpu_partoutput, flexgpu_readpartpositions, &
            flexgpu_concoutput, flexgpu_abi_sizes, flexgpu_comm_init_host, flexgpu_count_particles, flexgpu_set_option, flexgpu_get_info, &
            flexgpu_release_init, flexgpu_releaseparticles, flexgpu_split_particles, flexgpu_calcpar, &
            flexgpu_redist_plan, flexgpu_redist_bytes, flexgpu_redist_pack, flexgpu_redist_unpack, &
            flexgpu_checkpoint_write, flexgpu_checkpoint_read, &
            flexgpu_conv_init, flexgpu_upload_conv_fields, flexgpu_convmix, flexgpu_cbaseflux
#ifdef FLEXGPU_NESTS
  public :: flexgpu_upload_nests, flexgpu_upload_wet_nest_fields, flexgpu_nests_init, flexgpu_verttransform_nests, &
            flexgpu_upload_conv_nest_fields, flexgpu_cbaseflux_nests, flexgpu_upload_diag_nest_fields
#endif

  integer, parameter :: FPX_MAXSPEC = 5

  type, bind(C) :: fpx_config
    integer(c_int32_t) :: struct_bytes, device, compute_real_bytes, host_real_bytes
    integer(c_int64_t) :: max_particles
    integer(c_int32_t) :: nx, ny, nz, nmixz
    integer(c_int32_t) :: nxmax, nymax, nzmax
    real(c_double) :: dx, dy, xlon0, ylat0
    integer(c_int32_t) :: xglobal, nglobal, sglobal
    real(c_double) :: switchnorthg, switchsouthg
    real(c_double) :: northpolemap(9), southpolemap(9)
    integer(c_int32_t) :: ldirect, lsynctime, method, mintime, ifine, turbswitch, cblflag, mdomainfill, lsettling
    real(c_double) :: ctl
    real(c_double) :: d_trop, d_strat, turbmesoscale
    integer(c_int32_t) :: nspec, maxspec
    integer(c_int32_t) :: drydep, drydepspec(FPX_MAXSPEC)
    real(c_double) :: density(FPX_MAXSPEC), dquer(FPX_MAXSPEC), vsetaver(FPX_MAXSPEC), cunningham(FPX_MAXSPEC)
    real(c_double) :: decay(FPX_MAXSPEC)
    integer(c_int32_t) :: mquasilag
    integer(c_int32_t) :: lage_last
    integer(c_int32_t) :: rng_mode
    integer(c_int64_t) :: seed
    integer(c_int32_t) :: sort_interval
    integer(c_int32_t) :: par_nxmax
    integer(c_int64_t) :: particle_base
    integer(c_int32_t) :: drybkdep, wetbkdep
    integer(c_int32_t) :: turboff, interpolhmix
    integer(c_int32_t) :: blend_mode
    integer(c_int32_t) :: pbl_slice_passes
    integer(c_int64_t) :: global_particles
    integer(c_int32_t) :: ipout, iflux, linit_cond
    integer(c_int32_t) :: reserved(3)
  end type fpx_config

  type, bind(C) :: fpx_fields
    type(c_ptr) :: uu, vv, ww, uupol, vvpol, rho, drhodz, tt
    type(c_ptr) :: hmix, ustar, wstar, oli, tropopause
    type(c_ptr) :: vdep
  end type fpx_fields

  type, bind(C) :: fpx_model_levels
    type(c_ptr) :: uuh, vvh, pvh, wwh, tth, qvh, ps, tt2, td2, akz, bkz, aknew, bknew
    integer(c_int32_t) :: nuvz, nwz, init, pin_host
    real(c_double) :: nest_dy, nest_ylat0
  end type fpx_model_levels

  type, bind(C) :: fpx_fields_out
    type(c_ptr) :: uu, vv, ww, tt, qv, pv, rho, drhodz, uupol, vvpol
    type(c_ptr) :: height
    type(c_ptr) :: nmixz
  end type fpx_fields_out

  type, bind(C) :: fpx_diag_fields
    type(c_ptr) :: oro, pv, qv, tt
  end type fpx_diag_fields

  type, bind(C) :: fpx_restart
    real(c_double) :: jul_header, bdate
    integer(c_int32_t) :: mintime, nclassunc, itrasplit, reserved
  end type fpx_restart

  type, bind(C) :: fpx_concout
    type(c_ptr) :: area, volume
    real(c_double) :: outnum
    integer(c_int32_t) :: wetdep, drydep, nest, iout
    type(c_ptr) :: prefix_pptv, outheight
    real(c_double) :: outlon0, outlat0
    real(c_double) :: weightmolar(FPX_MAXSPEC)
    integer(c_int32_t) :: reduced, reserved
  end type fpx_concout

  integer, parameter :: FPX_MAXNESTS = 4
  type, bind(C) :: fpx_nests
    integer(c_int32_t) :: struct_bytes, numbnests, nxmaxn, nymaxn
    integer(c_int32_t) :: nxn(FPX_MAXNESTS), nyn(FPX_MAXNESTS)
    real(c_double) :: xln(FPX_MAXNESTS), yln(FPX_MAXNESTS), xrn(FPX_MAXNESTS), yrn(FPX_MAXNESTS)
    real(c_double) :: xresoln(FPX_MAXNESTS), yresoln(FPX_MAXNESTS)
  end type fpx_nests

  type, bind(C) :: fpx_particles
    type(c_ptr) :: xtra1, ytra1
    type(c_ptr) :: ztra1, uap, ucp, uzp, us, vs, ws
    type(c_ptr) :: itra1, itramem, idt, npoint, nclass
    type(c_ptr) :: cbt
    type(c_ptr) :: xmass1
    integer(c_int64_t) :: xmass1_ld
    type(c_ptr) :: itrasplit
    type(c_ptr) :: xscav_frac1
  end type fpx_particles

  type, bind(C) :: fpx_calcpar_in
    type(c_ptr) :: surfstr, sshf, akm, bkm, excessoro, vdep
    integer(c_int32_t) :: lsubgrid
    integer(c_int32_t) :: reserved(3)
  end type fpx_calcpar_in
  type, bind(C) :: fpx_calcpar_out
    type(c_ptr) :: ustar, wstar, oli, hmix, tropopause
  end type fpx_calcpar_out
  type, bind(C) :: fpx_conv_config
    integer(c_int32_t) :: struct_bytes, nuvz, nconvlev, reserved
    type(c_ptr) :: akz, bkz, akm, bkm
  end type fpx_conv_config
  type, bind(C) :: fpx_conv_fields
    type(c_ptr) :: ps, tt2, td2, tth, qvh
    integer(c_int32_t) :: nuvzmax, reserved
  end type fpx_conv_fields

  type, bind(C) :: fpx_release
    integer(c_int32_t) :: struct_bytes, numpoint
    type(c_ptr) :: ireleasestart, ireleaseend, kindz
    type(c_ptr) :: xpoint1, xpoint2, ypoint1, ypoint2, zpoint1, zpoint2
    type(c_ptr) :: point_hour, area_hour, point_dow, area_dow
    real(c_double) :: bdate
    integer(c_int32_t) :: itsplit, ind_rel, nclassunc
    integer(c_int32_t) :: reserved(5)
  end type fpx_release

  type, bind(C) :: fpx_step_stats
    integer(c_int64_t) :: n_due, n_initialized, n_left_domain, n_min_mass, n_max_age
    integer(c_int64_t) :: nan_count, nan_count2, n_bad_position
    real(c_double) :: kernel_ms
  end type fpx_step_stats

  interface
    integer(c_int) function fpx_create(h, cfg) bind(C, name='fpx_create')
      import :: c_ptr, c_int, fpx_config
      type(c_ptr), intent(out) :: h
      type(fpx_config), intent(in) :: cfg
    end function
    integer(c_int) function fpx_destroy(h) bind(C, name='fpx_destroy')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function fpx_calcpar(h, slot, c, o) bind(C, name='fpx_calcpar')
      import :: c_ptr, c_int, c_int32_t, fpx_calcpar_in, fpx_calcpar_out
      type(c_ptr), value :: h
      integer(c_int32_t), value :: slot
      type(fpx_calcpar_in), intent(in) :: c
      type(fpx_calcpar_out), intent(in) :: o
    end function
    integer(c_int) function fpx_release_init(h, r) bind(C, name='fpx_release_init')
      import :: c_ptr, c_int, fpx_release
      type(c_ptr), value :: h
      type(fpx_release), intent(in) :: r
    end function
    integer(c_int) function fpx_releaseparticles(h, itime, numpart, npc, xmasssave, rho_rel, nreleased) bind(C, name='fpx_releaseparticles')
      import :: c_ptr, c_int, c_int32_t, c_int64_t
      type(c_ptr), value :: h, xmasssave, rho_rel, nreleased
      integer(c_int32_t), value :: itime
      integer(c_int64_t), intent(inout) :: numpart
      integer(c_int32_t), intent(inout) :: npc
    end function
    integer(c_int) function fpx_split_particles(h, itime, numpart) bind(C, name='fpx_split_particles')
      import :: c_ptr, c_int, c_int32_t, c_int64_t
      type(c_ptr), value :: h
      integer(c_int32_t), value :: itime
      integer(c_int64_t), intent(inout) :: numpart
    end function
    integer(c_int) function fpx_redist_plan(npart, nranks, rank, ipout, role, peer, num_trans) bind(C, name='fpx_redist_plan')
      import :: c_int, c_int32_t, c_int64_t
      integer(c_int64_t), intent(in) :: npart(*)
      integer(c_int32_t), value :: nranks, rank, ipout
      integer(c_int32_t), intent(out) :: role, peer
      integer(c_int64_t), intent(out) :: num_trans
    end function
    integer(c_int64_t) function fpx_redist_bytes(h, num_trans) bind(C, name='fpx_redist_bytes')
      import :: c_ptr, c_int64_t
      type(c_ptr), value :: h
      integer(c_int64_t), value :: num_trans
    end function
    integer(c_int) function fpx_redist_pack(h, itime, num_trans, buf, buf_bytes, numpart) bind(C, name='fpx_redist_pack')
      import :: c_ptr, c_int, c_int32_t, c_int64_t
      type(c_ptr), value :: h, buf
      integer(c_int32_t), value :: itime
      integer(c_int64_t), value :: num_trans, buf_bytes
      integer(c_int64_t), intent(inout) :: numpart
    end function
    integer(c_int) function fpx_redist_unpack(h, itime, num_trans, buf, buf_bytes, numpart) bind(C, name='fpx_redist_unpack')
      import :: c_ptr, c_int, c_int32_t, c_int64_t
      type(c_ptr), value :: h, buf
      integer(c_int32_t), value :: itime
      integer(c_int64_t), value :: num_trans, buf_bytes
      integer(c_int64_t), intent(inout) :: numpart
    end function
    integer(c_int) function fpx_set_release_heights(h, numpoint, z1, z2) bind(C, name='fpx_set_release_heights')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h, z1, z2
      integer(c_int32_t), value :: numpoint
    end function
    integer(c_int) function fpx_set_release_points(h, numpoint, xmass, npart) bind(C, name='fpx_set_release_points')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h, xmass, npart
      integer(c_int32_t), value :: numpoint
    end function
    type(c_ptr) function fpx_last_error() bind(C, name='fpx_last_error')
      import :: c_ptr
    end function
    integer(c_int) function fpx_set_height(h, height, n) bind(C, name='fpx_set_height')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h, height
      integer(c_int32_t), value :: n
    end function
    integer(c_int) function fpx_upload_fields(h, slot, f) bind(C, name='fpx_upload_fields')
      import :: c_ptr, c_int, c_int32_t, fpx_fields
      type(c_ptr), value :: h
      integer(c_int32_t), value :: slot
      type(fpx_fields), intent(in) :: f
    end function
    integer(c_int) function fpx_verttransform_ecmwf(h, slot, m, sfc, o) bind(C, name='fpx_verttransform_ecmwf')
      import :: c_ptr, c_int, c_int32_t, fpx_model_levels, fpx_fields, fpx_fields_out
      type(c_ptr), value :: h
      integer(c_int32_t), value :: slot
      type(fpx_model_levels), intent(in) :: m
      type(c_ptr), value :: sfc            ! c_loc of a type(fpx_fields), or c_null_ptr: flexgpu_calcpar follows
      type(fpx_fields_out), intent(in) :: o
    end function fpx_verttransform_ecmwf
    integer(c_int) function fpx_upload_diag_fields(h, slot, f) bind(C, name='fpx_upload_diag_fields')
      import :: c_ptr, c_int, c_int32_t, fpx_diag_fields
      type(c_ptr), value :: h
      integer(c_int32_t), value :: slot
      type(fpx_diag_fields), intent(in) :: f
    end function fpx_upload_diag_fields
    integer(c_int) function fpx_partoutput(h, itime, path, nrec) bind(C, name='fpx_partoutput')
      import :: c_ptr, c_int, c_int32_t, c_int64_t, c_char
      type(c_ptr), value :: h
      integer(c_int32_t), value :: itime
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int64_t), intent(out) :: nrec
    end function fpx_partoutput
    integer(c_int) function fpx_conv_init(h, c) bind(C, name='fpx_conv_init')
      import :: c_ptr, c_int, fpx_conv_config
      type(c_ptr), value :: h
      type(fpx_conv_config), intent(in) :: c
    end function fpx_conv_init
    integer(c_int) function fpx_upload_conv_fields(h, slot, f) bind(C, name='fpx_upload_conv_fields')
      import :: c_ptr, c_int, c_int32_t, fpx_conv_fields
      type(c_ptr), value :: h
      integer(c_int32_t), value :: slot
      type(fpx_conv_fields), intent(in) :: f
    end function fpx_upload_conv_fields
    integer(c_int) function fpx_convmix(h, itime, nmoved) bind(C, name='fpx_convmix')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h
      integer(c_int32_t), value :: itime
      type(c_ptr), value :: nmoved
    end function fpx_convmix
    integer(c_int) function fpx_upload_diag_nest_fields(h, nest, slot, f) bind(C, name='fpx_upload_diag_nest_fields')
      import :: c_ptr, c_int, c_int32_t, fpx_diag_fields
      type(c_ptr), value :: h
      integer(c_int32_t), value :: nest, slot
      type(fpx_diag_fields), intent(in) :: f
    end function fpx_upload_diag_nest_fields
    integer(c_int) function fpx_upload_conv_nest_fields(h, nest, slot, f) bind(C, name='fpx_upload_conv_nest_fields')
      import :: c_ptr, c_int, c_int32_t, fpx_conv_fields
      type(c_ptr), value :: h
      integer(c_int32_t), value :: nest, slot
      type(fpx_conv_fields), intent(in) :: f
    end function fpx_upload_conv_nest_fields
    integer(c_int) function fpx_get_cbaseflux_nest(h, nest, cb) bind(C, name='fpx_get_cbaseflux_nest')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h, cb
      integer(c_int32_t), value :: nest
    end function fpx_get_cbaseflux_nest
    integer(c_int) function fpx_set_cbaseflux_nest(h, nest, cb) bind(C, name='fpx_set_cbaseflux_nest')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h, cb
      integer(c_int32_t), value :: nest
    end function fpx_set_cbaseflux_nest
    integer(c_int) function fpx_get_cbaseflux(h, cb) bind(C, name='fpx_get_cbaseflux')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, cb
    end function fpx_get_cbaseflux
    integer(c_int) function fpx_set_cbaseflux(h, cb) bind(C, name='fpx_set_cbaseflux')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, cb
    end function fpx_set_cbaseflux
    integer(c_int) function fpx_checkpoint_write(h, path, itime, npc) bind(C, name='fpx_checkpoint_write')
      import :: c_ptr, c_int, c_int32_t, c_char
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int32_t), value :: itime, npc
    end function fpx_checkpoint_write
    integer(c_int) function fpx_checkpoint_read(h, path, itime, np, npc) bind(C, name='fpx_checkpoint_read')
      import :: c_ptr, c_int, c_int32_t, c_int64_t, c_char
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int32_t), intent(out) :: itime, npc
      integer(c_int64_t), intent(out) :: np
    end function fpx_checkpoint_read
    integer(c_int) function fpx_readpartpositions(h, path, r, np, npc, itimein) bind(C, name='fpx_readpartpositions')
      import :: c_ptr, c_int, c_int32_t, c_int64_t, c_char, fpx_restart
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: path(*)
      type(fpx_restart), intent(in) :: r
      integer(c_int64_t), intent(out) :: np
      integer(c_int32_t), intent(out) :: npc, itimein
    end function fpx_readpartpositions
    integer(c_int) function fpx_verttransform_nest(h, nest, slot, m, sfc, o) bind(C, name='fpx_verttransform_nest')
      import :: c_ptr, c_int, c_int32_t, fpx_model_levels, fpx_fields, fpx_fields_out
      type(c_ptr), value :: h
      integer(c_int32_t), value :: nest, slot
      type(fpx_model_levels), intent(in) :: m
      type(fpx_fields), intent(in) :: sfc
      type(fpx_fields_out), intent(in) :: o
    end function fpx_verttransform_nest
    integer(c_int) function fpx_concoutput(h, itime, c, prefix, clear) bind(C, name='fpx_concoutput')
      import :: c_ptr, c_int, c_int32_t, c_char, fpx_concout
      type(c_ptr), value :: h
      integer(c_int32_t), value :: itime, clear
      type(fpx_concout), intent(in) :: c
      character(kind=c_char), intent(in) :: prefix(*)
    end function fpx_concoutput
    integer(c_int) function fpx_nests_init(h, n) bind(C, name='fpx_nests_init')
      import :: c_ptr, c_int, fpx_nests
      type(c_ptr), value :: h
      type(fpx_nests), intent(in) :: n
    end function
    integer(c_int) function fpx_upload_nest_fields(h, nest, slot, f) bind(C, name='fpx_upload_nest_fields')
      import :: c_ptr, c_int, c_int32_t, fpx_fields
      type(c_ptr), value :: h
      integer(c_int32_t), value :: nest, slot
      type(fpx_fields), intent(in) :: f
    end function
    integer(c_int) function fpx_set_windtime(h, mt, mi) bind(C, name='fpx_set_windtime')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h
      integer(c_int32_t), intent(in) :: mt(2), mi(2)
    end function
    integer(c_int) function fpx_rng_set_table(h, tab, n) bind(C, name='fpx_rng_set_table')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h, tab
      integer(c_int32_t), value :: n
    end function
    integer(c_int) function fpx_upload_particles(h, first, count, p) bind(C, name='fpx_upload_particles')
      import :: c_ptr, c_int, c_int64_t, fpx_particles
      type(c_ptr), value :: h
      integer(c_int64_t), value :: first, count
      type(fpx_particles), intent(in) :: p
    end function
    integer(c_int) function fpx_download_particles(h, first, count, p) bind(C, name='fpx_download_particles')
      import :: c_ptr, c_int, c_int64_t, fpx_particles
      type(c_ptr), value :: h
      integer(c_int64_t), value :: first, count
      type(fpx_particles), intent(in) :: p
    end function
    integer(c_int) function fpx_step(h, itime, st) bind(C, name='fpx_step')
      import :: c_ptr, c_int, c_int32_t, fpx_step_stats
      type(c_ptr), value :: h
      integer(c_int32_t), value :: itime
      type(fpx_step_stats), intent(out) :: st
    end function
  end interface

  integer, parameter :: FPX_MAXAGECLASS = 8
  type, bind(C) :: fpx_outgrid
    integer(c_int32_t) :: struct_bytes
    integer(c_int32_t) :: numxgrid, numygrid, numzgrid
    real(c_double) :: dxout, dyout, xoutshift, youtshift
    integer(c_int32_t) :: maxpointspec_act, nclassunc, nageclass
    integer(c_int32_t) :: lage(FPX_MAXAGECLASS)
    integer(c_int32_t) :: ind_samp, ioutputforeachrelease
    integer(c_int32_t) :: lusekerneloutput
    integer(c_int32_t) :: reserved(5)
  end type
  type, bind(C) :: fpx_outgrid_nest
    integer(c_int32_t) :: struct_bytes
    integer(c_int32_t) :: numxgridn, numygridn
    real(c_double) :: dxoutn, dyoutn, xoutshiftn, youtshiftn
    integer(c_int32_t) :: reserved(4)
  end type
  type, bind(C) :: fpx_wet_config
    integer(c_int32_t) :: struct_bytes
    integer(c_int32_t) :: wetdepspec(FPX_MAXSPEC)
    real(c_double) :: weta_gas(FPX_MAXSPEC), wetb_gas(FPX_MAXSPEC)
    real(c_double) :: crain_aero(FPX_MAXSPEC), csnow_aero(FPX_MAXSPEC)
    real(c_double) :: ccn_aero(FPX_MAXSPEC), in_aero(FPX_MAXSPEC)
    real(c_double) :: henry(FPX_MAXSPEC)
    integer(c_int32_t) :: readclouds
    integer(c_int32_t) :: reserved(7)
  end type
  type, bind(C) :: fpx_wet_fields
    type(c_ptr) :: lsprec, convprec, tcc, ctwc, tt
    type(c_ptr) :: clouds, cloudsh
  end type

  interface
    integer(c_int) function fpx_outgrid_init(h, g, oh) bind(C, name='fpx_outgrid_init')
      import :: c_ptr, c_int, fpx_outgrid
      type(c_ptr), value :: h
      type(fpx_outgrid), intent(in) :: g
      type(c_ptr), value :: oh
    end function
    integer(c_int) function fpx_outgrid_nest_init(h, g) bind(C, name='fpx_outgrid_nest_init')
      import :: c_ptr, c_int, fpx_outgrid_nest
      type(c_ptr), value :: h
      type(fpx_outgrid_nest), intent(in) :: g
    end function
    integer(c_int) function fpx_receptors_init(h, n, x, y, a) bind(C, name='fpx_receptors_init')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h
      integer(c_int32_t), value :: n
      type(c_ptr), value :: x, y, a
    end function
    integer(c_int) function fpx_set_output_times(h, loutnext, loutstep) bind(C, name='fpx_set_output_times')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h
      integer(c_int32_t), value :: loutnext, loutstep
    end function
    integer(c_int) function fpx_conccalc(h, itime, weight) bind(C, name='fpx_conccalc')
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: h
      integer(c_int32_t), value :: itime
      real(c_double), value :: weight
    end function
    integer(c_int) function fpx_get_grids(h, g, d, allreduce, clear) bind(C, name='fpx_get_grids')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h, g, d
      integer(c_int32_t), value :: allreduce, clear
    end function
    integer(c_int) function fpx_comm_init_host(h, nranks, rank, fn, user) bind(C, name='fpx_comm_init_host')
      import :: c_ptr, c_funptr, c_int, c_int32_t
      type(c_ptr), value :: h, user
      type(c_funptr), value :: fn
      integer(c_int32_t), value :: nranks, rank
    end function
    integer(c_int) function fpx_count_particles(h, local, total, allreduce) bind(C, name='fpx_count_particles')
      import :: c_ptr, c_int, c_int32_t, c_int64_t
      type(c_ptr), value :: h
      integer(c_int64_t), intent(out) :: local(2), total(2)
      integer(c_int32_t), value :: allreduce
    end function
    integer(c_int) function fpx_set_option(h, name, value) bind(C, name='fpx_set_option')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: name(*), value(*)
    end function
    integer(c_int) function fpx_get_info(h, name, value) bind(C, name='fpx_get_info')
      import :: c_ptr, c_int, c_char, c_int64_t
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int64_t), intent(out) :: value
    end function
    integer(c_int) function fpx_get_wetgrid(h, w, allreduce) bind(C, name='fpx_get_wetgrid')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h, w
      integer(c_int32_t), value :: allreduce
    end function
    integer(c_int) function fpx_get_grids_nest(h, g, d, w, allreduce, clear) bind(C, name='fpx_get_grids_nest')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h, g, d, w
      integer(c_int32_t), value :: allreduce, clear
    end function
    integer(c_int) function fpx_get_receptors(h, c, ld, allreduce, clear) bind(C, name='fpx_get_receptors')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h, c
      integer(c_int32_t), value :: ld, allreduce, clear
    end function
    integer(c_int) function fpx_wet_init(h, w) bind(C, name='fpx_wet_init')
      import :: c_ptr, c_int, fpx_wet_config
      type(c_ptr), value :: h
      type(fpx_wet_config), intent(in) :: w
    end function
    integer(c_int) function fpx_upload_wet_fields(h, slot, f) bind(C, name='fpx_upload_wet_fields')
      import :: c_ptr, c_int, c_int32_t, fpx_wet_fields
      type(c_ptr), value :: h
      integer(c_int32_t), value :: slot
      type(fpx_wet_fields), intent(in) :: f
    end function
    integer(c_int) function fpx_upload_wet_nest_fields(h, nest, slot, f, rc) bind(C, name='fpx_upload_wet_nest_fields')
      import :: c_ptr, c_int, c_int32_t, fpx_wet_fields
      type(c_ptr), value :: h
      integer(c_int32_t), value :: nest, slot, rc
      type(fpx_wet_fields), intent(in) :: f
    end function
    integer(c_int) function fpx_wetdepo(h, itime, ltsample, loutnext) bind(C, name='fpx_wetdepo')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h
      integer(c_int32_t), value :: itime, ltsample, loutnext
    end function
  end interface

  type(c_ptr), save :: flexgpu_handle = c_null_ptr

contains

  ! Address of a host array element.  com_mod's arrays carry no TARGET attribute, so c_loc is
  ! applied to a TARGET dummy instead (sequence association: no copy is made for these
  ! contiguous arrays, and the engine only uses the address during the call).
  function loc_r(x) result(p)
    real, target, intent(in) :: x(*)
    type(c_ptr) :: p
    p = c_loc(x)
  end function loc_r
  function loc_d(x) result(p)
    real(kind=dp), target, intent(in) :: x(*)
    type(c_ptr) :: p
    p = c_loc(x)
  end function loc_d
  function loc_i(x) result(p)
    integer, target, intent(in) :: x(*)
    type(c_ptr) :: p
    p = c_loc(x)
  end function loc_i
  function loc_i2(x) result(p)
    integer(kind=2), target, intent(in) :: x(*)
    type(c_ptr) :: p
    p = c_loc(x)
  end function loc_i2

  ! c_sizeof of every bind(C) type of this module, in the order config, fields, particles, step_stats, model_levels,
  ! fields_out, diag_fields, restart, concout, nests -- compared with the C compiler's sizeof by the tests
  subroutine flexgpu_abi_sizes(sizes)
    integer, intent(out) :: sizes(10)
    type(fpx_config) :: a
    type(fpx_fields) :: b
    type(fpx_particles) :: c
    type(fpx_step_stats) :: d
    type(fpx_model_levels) :: e
    type(fpx_fields_out) :: f
    type(fpx_diag_fields) :: g
    type(fpx_restart) :: h
    type(fpx_concout) :: i
    type(fpx_nests) :: j
    sizes = (/ int(c_sizeof(a)), int(c_sizeof(b)), int(c_sizeof(c)), int(c_sizeof(d)), int(c_sizeof(e)), &
               int(c_sizeof(f)), int(c_sizeof(g)), int(c_sizeof(h)), int(c_sizeof(i)), int(c_sizeof(j)) /)
  end subroutine flexgpu_abi_sizes

  subroutine flexgpu_last_error(msg)
    character(len=*), intent(out) :: msg
    type(c_ptr) :: p
    character(kind=c_char), pointer :: s(:)
    integer :: i
    msg = ''
    p = fpx_last_error()
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, [len(msg)])
    do i = 1, len(msg)
      if (s(i) == c_null_char) exit
      msg(i:i) = s(i)
    end do
  end subroutine flexgpu_last_error

  ! com_mod/par_mod -> fpx_config; creates the engine on `device` for `nmaxpart` particles.
  ! defer_height: the z levels do not exist yet -- the first flexgpu_verttransform derives them
  ! particle_base (MPI host): global number of this rank's first particle, so that the counter RNG does not depend on the
  ! number of ranks; global_particles (MPI host): maxpart of the whole run = the sum over the ranks -- decisions every rank
  ! must take alike (the time-blended wind packs) are taken from it, never from what one rank holds; blend_mode: 0 from
  ! global_particles, 1 on, 2 off.
  ! The run's ipout / iflux / linit_cond travel along: the engine refuses (ierr = -5) a run whose particle loop would also
  ! call partpos_average, calcfluxes or initial_cond_calc (timemanager.f90:617,623,631,702), which it does not compute.
  ! turboff / interpolhmix: the host's compile-time parameters (com_mod.f90:777-778) become the engine's run-time switches.
  subroutine flexgpu_init(ierr, device, nmaxpart, compute_real_bytes, rng_mode, seed, defer_height, particle_base, &
                          global_particles, blend_mode)
    integer, intent(out) :: ierr
    integer, intent(in), optional :: device, nmaxpart, compute_real_bytes, rng_mode, blend_mode
    logical, intent(in), optional :: defer_height
    integer(c_int64_t), intent(in), optional :: seed, particle_base, global_particles
    type(fpx_config) :: cfg
    integer :: ks
    cfg%struct_bytes = int(c_sizeof(cfg), c_int32_t)
    cfg%device = 0; if (present(device)) cfg%device = device
    cfg%host_real_bytes = storage_size(1.0) / 8          ! 4 as shipped, 8 with -fdefault-real-8
    cfg%compute_real_bytes = 8; if (present(compute_real_bytes)) cfg%compute_real_bytes = compute_real_bytes
    cfg%max_particles = size(xtra1); if (present(nmaxpart)) cfg%max_particles = nmaxpart
    cfg%nx = nx; cfg%ny = ny; cfg%nz = nz; cfg%nmixz = nmixz
    cfg%nxmax = nxmax; cfg%nymax = nymax; cfg%nzmax = nzmax
    cfg%dx = dx; cfg%dy = dy; cfg%xlon0 = xlon0; cfg%ylat0 = ylat0
    cfg%xglobal = merge(1, 0, xglobal); cfg%nglobal = merge(1, 0, nglobal); cfg%sglobal = merge(1, 0, sglobal)
    cfg%switchnorthg = switchnorthg; cfg%switchsouthg = switchsouthg
    cfg%northpolemap = northpolemap; cfg%southpolemap = southpolemap
    cfg%ldirect = ldirect; cfg%lsynctime = lsynctime; cfg%method = method; cfg%mintime = mintime
    cfg%ifine = ifine; cfg%turbswitch = merge(1, 0, turbswitch); cfg%cblflag = cblflag
    cfg%mdomainfill = mdomainfill; cfg%lsettling = merge(1, 0, lsettling)
    cfg%ctl = ctl
    cfg%d_trop = d_trop; cfg%d_strat = d_strat; cfg%turbmesoscale = turbmesoscale
    cfg%nspec = nspec; cfg%maxspec = maxspec
    cfg%drydep = merge(1, 0, DRYDEP)
    cfg%drydepspec = 0; cfg%density = 0; cfg%dquer = 0; cfg%vsetaver = 0; cfg%cunningham = 1
    cfg%decay = 0
    do ks = 1, nspec
      cfg%drydepspec(ks) = merge(1, 0, DRYDEPSPEC(ks))
      cfg%density(ks) = density(ks); cfg%dquer(ks) = dquer(ks); cfg%vsetaver(ks) = vsetaver(ks)
      cfg%cunningham(ks) = cunningham(ks); cfg%decay(ks) = decay(ks)
    end do
    cfg%mquasilag = mquasilag
    cfg%lage_last = lage(nageclass)
    cfg%rng_mode = 0; if (present(rng_mode)) cfg%rng_mode = rng_mode
    cfg%seed = 24301_c_int64_t; if (present(seed)) cfg%seed = seed
    cfg%sort_interval = 8
    cfg%par_nxmax = nxmax          ! eps = nxmax/3.e5, advance.f90:107
    cfg%particle_base = 0; if (present(particle_base)) cfg%particle_base = particle_base
    cfg%drybkdep = merge(1, 0, DRYBKDEP); cfg%wetbkdep = merge(1, 0, WETBKDEP)   ! timemanager.f90:564-598 runs in flexgpu_step
    cfg%turboff = merge(1, 0, turboff); cfg%interpolhmix = merge(1, 0, interpolhmix)   ! com_mod.f90:777-778
    cfg%blend_mode = 0; if (present(blend_mode)) cfg%blend_mode = blend_mode
    cfg%pbl_slice_passes = 0
    cfg%global_particles = cfg%max_particles; if (present(global_particles)) cfg%global_particles = global_particles
    cfg%ipout = ipout; cfg%iflux = iflux; cfg%linit_cond = linit_cond
    cfg%reserved = 0
    ierr = fpx_create(flexgpu_handle, cfg)
    if (ierr /= 0) return
    ! point_mod xmass(numpoint,maxspec), npart(numpoint): indexed by npoint(j) in the particle loop
    ! (timemanager.f90:663-666, advance.f90:518-531)
    if (allocated(xmass) .and. allocated(npart)) then
      ierr = fpx_set_release_points(flexgpu_handle, int(size(npart), c_int32_t), loc_r(xmass), loc_i(npart))
      if (ierr /= 0) return
    end if
    if (WETBKDEP .and. allocated(zpoint1) .and. allocated(zpoint2)) then      ! timemanager.f90:590-591
      ierr = fpx_set_release_heights(flexgpu_handle, int(size(zpoint1), c_int32_t), loc_r(zpoint1), loc_r(zpoint2))
      if (ierr /= 0) return
    end if
    if (present(defer_height)) then
      if (defer_height) return
    end if
    ierr = fpx_set_height(flexgpu_handle, loc_r(height), int(nz, c_int32_t))
  end subroutine flexgpu_init

  subroutine flexgpu_finalize()
    integer :: ierr
    ierr = fpx_destroy(flexgpu_handle)
    flexgpu_handle = c_null_ptr
  end subroutine flexgpu_finalize

  ! bit-parity mode: hand the host's own rannumb table (FLEXPART.f90:56-59) to the engine
  subroutine flexgpu_use_table_rng(ierr)
    integer, intent(out) :: ierr
    ierr = fpx_rng_set_table(flexgpu_handle, loc_r(rannumb), int(maxrand, c_int32_t))
  end subroutine flexgpu_use_table_rng

  ! Replaces `call verttransform_ecmwf(n,uuh,vvh,wwh,pvh)` (getfields.f90:129,164,180) AND the upload of
  ! slot n: the eta -> z transform runs on the device and lands in the engine's gather layout.
  ! writeback (default .true.): uu, vv, ww, tt, qv, pv, rho, drhodz, uupol, vvpol of slot n are also
  ! copied back into com_mod for the host routines that still read them (partoutput, convmix, the
  ! cloud diagnostics); height(:) and nmixz are set on the first call as the reference does.
  ! pin_host = .true.: uuh ... td2 keep their addresses for the whole run (static arrays): registered once for DMA.
  ! device_calcpar = .true.: the 2-D fields of calcpar are not taken from com_mod; flexgpu_calcpar(n, ierr) computes them.
  subroutine flexgpu_verttransform(n, uuh, vvh, wwh, pvh, ierr, writeback, pin_host, device_calcpar)
    integer, intent(in) :: n
    real, intent(in) :: uuh(0:nxmax-1,0:nymax-1,nuvzmax), vvh(0:nxmax-1,0:nymax-1,nuvzmax)
    real, intent(in) :: pvh(0:nxmax-1,0:nymax-1,nuvzmax), wwh(0:nxmax-1,0:nymax-1,nwzmax)
    integer, intent(out) :: ierr
    logical, intent(in), optional :: writeback, pin_host, device_calcpar
    logical, save :: first = .true.
    type(fpx_model_levels) :: m
    type(fpx_fields), target :: f
    type(fpx_fields_out) :: o
    type(c_ptr) :: fp
    integer(c_int32_t), target :: nmixz_c
    logical :: wb
    wb = .true.; if (present(writeback)) wb = writeback
    m%uuh = loc_r(uuh); m%vvh = loc_r(vvh); m%pvh = loc_r(pvh); m%wwh = loc_r(wwh)
    m%tth = loc_r(tth(0,0,1,n)); m%qvh = loc_r(qvh(0,0,1,n))
    m%ps = loc_r(ps(0,0,1,n)); m%tt2 = loc_r(tt2(0,0,1,n)); m%td2 = loc_r(td2(0,0,1,n))
    m%akz = loc_r(akz); m%bkz = loc_r(bkz); m%aknew = loc_r(aknew); m%bknew = loc_r(bknew)
    m%nuvz = nuvz; m%nwz = nwz; m%init = merge(1, 0, first)
    m%pin_host = 0; if (present(pin_host)) m%pin_host = merge(1, 0, pin_host)
    m%nest_dy = 0; m%nest_ylat0 = 0
    f%uu = c_null_ptr; f%vv = c_null_ptr; f%ww = c_null_ptr; f%uupol = c_null_ptr; f%vvpol = c_null_ptr
    f%rho = c_null_ptr; f%drhodz = c_null_ptr; f%tt = c_null_ptr
    f%hmix = loc_r(hmix(0,0,1,n)); f%ustar = loc_r(ustar(0,0,1,n)); f%wstar = loc_r(wstar(0,0,1,n))
    f%oli = loc_r(oli(0,0,1,n)); f%tropopause = loc_r(tropopause(0,0,1,n))
    f%vdep = loc_r(vdep(0,0,1,n))
    o%uu = c_null_ptr; o%vv = c_null_ptr; o%ww = c_null_ptr; o%tt = c_null_ptr; o%qv = c_null_ptr
    o%pv = c_null_ptr; o%rho = c_null_ptr; o%drhodz = c_null_ptr; o%uupol = c_null_ptr; o%vvpol = c_null_ptr
    if (wb) then
      o%uu = loc_r(uu(0,0,1,n)); o%vv = loc_r(vv(0,0,1,n)); o%ww = loc_r(ww(0,0,1,n))
      o%tt = loc_r(tt(0,0,1,n)); o%qv = loc_r(qv(0,0,1,n)); o%pv = loc_r(pv(0,0,1,n))
      o%rho = loc_r(rho(0,0,1,n)); o%drhodz = loc_r(drhodz(0,0,1,n))
      if (nglobal .or. sglobal) then
        o%uupol = loc_r(uupol(0,0,1,n)); o%vvpol = loc_r(vvpol(0,0,1,n))
      end if
    end if
    o%height = loc_r(height)
    nmixz_c = nmixz
    o%nmixz = c_loc(nmixz_c)
    fp = c_loc(f)
    if (present(device_calcpar)) then
      if (device_calcpar) fp = c_null_ptr
    end if
    ierr = fpx_verttransform_ecmwf(flexgpu_handle, int(n, c_int32_t), m, fp, o)
    if (ierr /= 0) return
    nmixz = nmixz_c
    first = .false.
  end subroutine flexgpu_verttransform

  ! oro (slot = 0) or pv, qv, tt of one time slot: the fields only partoutput reads.  Not needed for
  ! slots that went through flexgpu_verttransform (the device keeps its own pv, qv, tt then).
  subroutine flexgpu_upload_diag_fields(slot, ierr)
    integer, intent(in) :: slot
    integer, intent(out) :: ierr
    type(fpx_diag_fields) :: f
    f%oro = loc_r(oro); f%pv = c_null_ptr; f%qv = c_null_ptr; f%tt = c_null_ptr
    if (slot .ge. 1) then
      f%pv = loc_r(pv(0,0,1,slot)); f%qv = loc_r(qv(0,0,1,slot)); f%tt = loc_r(tt(0,0,1,slot))
    end if
    ierr = fpx_upload_diag_fields(flexgpu_handle, int(slot, c_int32_t), f)
  end subroutine flexgpu_upload_diag_fields

  ! Replaces `call partoutput(itime)` (timemanager.f90:454): same file name (partoutput.f90:63-86),
  ! same bytes; the particles stay on the device.
  subroutine flexgpu_partoutput(itime, ierr, nrecords)
    integer, intent(in) :: itime
    integer, intent(out) :: ierr
    integer(c_int64_t), intent(out), optional :: nrecords
    real(kind=dp) :: jul
    integer :: jjjjmmdd, ihmmss
    character :: adate*8, atime*6
    character(len=200) :: fname
    integer(c_int64_t) :: nrec
    jul = bdate + real(itime, kind=dp) / 86400._dp
    call caldate(jul, jjjjmmdd, ihmmss)
    write(adate, '(i8.8)') jjjjmmdd
    write(atime, '(i6.6)') ihmmss
    if (ipout .eq. 1 .or. ipout .eq. 3) then
      fname = path(2)(1:length(2)) // 'partposit_' // adate // atime
    else
      fname = path(2)(1:length(2)) // 'partposit_end'
    end if
    ierr = fpx_partoutput(flexgpu_handle, int(itime, c_int32_t), trim(fname) // c_null_char, nrec)
    if (present(nrecords)) nrecords = nrec
  end subroutine flexgpu_partoutput

  ! Replaces the second half of `call readpartpositions` (readpartpositions.f90:115-148): the host has read
  ! the `header` file (:59-113) and passes its date and time; the dump path(2)//'partposit_end' is parsed
  ! on the device.  Sets numpart, numparticlecount and itrasplit as the routine does.
  subroutine flexgpu_readpartpositions(ibdatein, ibtimein, ierr)
    integer, intent(in) :: ibdatein, ibtimein
    integer, intent(out) :: ierr
    type(fpx_restart) :: r
    integer(c_int64_t) :: np
    integer(c_int32_t) :: npc, itimein
    real(kind=dp) :: juldate
    r%jul_header = juldate(ibdatein, ibtimein)
    r%bdate = bdate
    r%mintime = mintime
    r%nclassunc = nclassunc
    r%itrasplit = ldirect * itsplit; r%reserved = 0
    ierr = fpx_readpartpositions(flexgpu_handle, path(2)(1:length(2)) // 'partposit_end' // c_null_char, r, np, npc, itimein)
    if (ierr /= 0) return
    numpart = int(np)
    numparticlecount = npc
    itrasplit(1:numpart) = ldirect * itsplit
  end subroutine flexgpu_readpartpositions

  ! ---- convective mixing on the device (SURVEY section 8 f3) -----------------------------------------------------------
  ! after gridcheck_ecmwf (nuvz, nconvlev, akz, bkz, akm, bkm): level structure; cbaseflux starts from zero as in the reference
  subroutine flexgpu_conv_init(ierr)
    integer, intent(out) :: ierr
    type(fpx_conv_config) :: c
    c%struct_bytes = int(c_sizeof(c), c_int32_t)
    c%nuvz = nuvz; c%nconvlev = nconvlev; c%reserved = 0
    c%akz = loc_r(akz); c%bkz = loc_r(bkz); c%akm = loc_r(akm); c%bkm = loc_r(bkm)
    ierr = fpx_conv_init(flexgpu_handle, c)
  end subroutine flexgpu_conv_init

  ! after readwind_ecmwf filled slot n = memind(k) (getfields.f90): ps, tt2, td2, tth, qvh of that slot
  subroutine flexgpu_upload_conv_fields(n, ierr)
    integer, intent(in) :: n
    integer, intent(out) :: ierr
    type(fpx_conv_fields) :: f
    f%ps = loc_r(ps(0,0,1,n)); f%tt2 = loc_r(tt2(0,0,1,n)); f%td2 = loc_r(td2(0,0,1,n))
    f%tth = loc_r(tth(0,0,1,n)); f%qvh = loc_r(qvh(0,0,1,n))
    f%nuvzmax = nuvzmax; f%reserved = 0
    ierr = fpx_upload_conv_fields(flexgpu_handle, int(n, c_int32_t), f)
  end subroutine flexgpu_upload_conv_fields

  ! replaces `call convmix(itime,metdata_format)` (timemanager.f90:258-262; backward runs :183-187)
  subroutine flexgpu_convmix(itime, ierr)
    integer, intent(in) :: itime
    integer, intent(out) :: ierr
    ierr = fpx_convmix(flexgpu_handle, int(itime, c_int32_t), c_null_ptr)
  end subroutine flexgpu_convmix

  ! conv_mod cbaseflux(0:nxmax-1,0:nymax-1) <-> the engine's field (set = .true.: host -> device, e.g. from a restart file)
  subroutine flexgpu_cbaseflux(set, ierr)
    logical, intent(in) :: set
    integer, intent(out) :: ierr
    real, allocatable, target :: buf(:,:)
    allocate(buf(0:nx-1,0:ny-1))
    if (set) then
      buf = cbaseflux(0:nx-1,0:ny-1)
      ierr = fpx_set_cbaseflux(flexgpu_handle, c_loc(buf))
    else
      ierr = fpx_get_cbaseflux(flexgpu_handle, c_loc(buf))
      if (ierr == 0) cbaseflux(0:nx-1,0:ny-1) = buf
    end if
    deallocate(buf)
  end subroutine flexgpu_cbaseflux

  ! Lossless restart file path(2)//'flexgpu_checkpoint' (no reference counterpart: partoutput / readpartpositions lose the
  ! turbulent state, DESIGN.md section 11): everything the particle loop carries.  A run continued with
  ! flexgpu_checkpoint_read is the uninterrupted run.
  subroutine flexgpu_checkpoint_write(itime, ierr)
    integer, intent(in) :: itime
    integer, intent(out) :: ierr
    ierr = fpx_checkpoint_write(flexgpu_handle, path(2)(1:length(2)) // 'flexgpu_checkpoint' // c_null_char, int(itime, c_int32_t), &
                                int(numparticlecount, c_int32_t))
  end subroutine flexgpu_checkpoint_write

  subroutine flexgpu_checkpoint_read(itime, ierr)
    integer, intent(out) :: itime, ierr
    integer(c_int64_t) :: np
    integer(c_int32_t) :: npc, it
    ierr = fpx_checkpoint_read(flexgpu_handle, path(2)(1:length(2)) // 'flexgpu_checkpoint' // c_null_char, it, np, npc)
    if (ierr /= 0) return
    itime = it
    numpart = int(np)
    numparticlecount = npc
  end subroutine flexgpu_checkpoint_read

  ! Writes the grid_conc_<date><time>_<species> files of `call concoutput(itime,outnum,...)` (timemanager.f90:384;
  ! forward runs, iout = 1 or 3 or 5) from the device's sampling grids and zeroes gridunc as the routine does.
  ! dates, grid_pptv_*, factor_drygrid and the receptor files remain with the host.  4-byte default real only.
  ! reduced = .true. (MPI host, after flexgpu_get_grids(clear, ierr, allreduce=.true.)): the files hold the sums over all
  ! ranks, as concoutput_mpi.f90 writes them from gridunc0 / drygridunc0 / wetgridunc0 on the root process.
  subroutine flexgpu_concoutput(itime, outnum, ierr, reduced)
    integer, intent(in) :: itime
    real, intent(in) :: outnum
    integer, intent(out) :: ierr
    logical, intent(in), optional :: reduced
    type(fpx_concout) :: c
    real(kind=dp) :: jul
    integer :: jjjjmmdd, ihmmss
    character :: adate*8, atime*6
    character(len=200) :: prefix
    jul = bdate + real(itime, kind=dp) / 86400._dp
    call caldate(jul, jjjjmmdd, ihmmss)
    write(adate, '(i8.8)') jjjjmmdd
    write(atime, '(i6.6)') ihmmss
    prefix = path(2)(1:length(2)) // 'grid_conc_' // adate // atime // '_'
    c%area = loc_r(area); c%volume = loc_r(volume)
    c%outnum = outnum
    c%wetdep = merge(1, 0, WETDEP); c%drydep = merge(1, 0, DRYDEP); c%nest = 0; c%iout = 1
    c%prefix_pptv = c_null_ptr; c%outheight = c_null_ptr; c%outlon0 = 0; c%outlat0 = 0; c%weightmolar = 1   ! grid_pptv_*: through the C ABI
    c%reduced = 0; c%reserved = 0
    if (present(reduced)) c%reduced = merge(1, 0, reduced)
    ierr = fpx_concoutput(flexgpu_handle, int(itime, c_int32_t), c, trim(prefix) // c_null_char, 1_c_int32_t)
  end subroutine flexgpu_concoutput

  ! one time slot of the com_mod fields (slot = the value found in memind(k))
  subroutine flexgpu_upload_fields(slot, ierr)
    integer, intent(in) :: slot
    integer, intent(out) :: ierr
    type(fpx_fields) :: f
    f%uu = loc_r(uu(0,0,1,slot)); f%vv = loc_r(vv(0,0,1,slot)); f%ww = loc_r(ww(0,0,1,slot))
    f%uupol = loc_r(uupol(0,0,1,slot)); f%vvpol = loc_r(vvpol(0,0,1,slot))
    f%rho = loc_r(rho(0,0,1,slot)); f%drhodz = loc_r(drhodz(0,0,1,slot)); f%tt = loc_r(tt(0,0,1,slot))
    f%hmix = loc_r(hmix(0,0,1,slot)); f%ustar = loc_r(ustar(0,0,1,slot)); f%wstar = loc_r(wstar(0,0,1,slot))
    f%oli = loc_r(oli(0,0,1,slot)); f%tropopause = loc_r(tropopause(0,0,1,slot))
    f%vdep = loc_r(vdep(0,0,1,slot))
    ierr = fpx_upload_fields(flexgpu_handle, int(slot, c_int32_t), f)
  end subroutine flexgpu_upload_fields

#ifdef FLEXGPU_NESTS
  ! nested grids: geometry (gridcheck_nests.f90:362-378) and both time slots of every nest.
  ! Compiled only against a par_mod with maxnests >= 1 (zero-sized nest arrays otherwise).
  ! geometry_only = .true.: only fpx_nests_init (the fields then come from flexgpu_verttransform_nests)
  subroutine flexgpu_upload_nests(ierr, geometry_only)
    integer, intent(out) :: ierr
    logical, intent(in), optional :: geometry_only
    type(fpx_nests) :: n
    type(fpx_fields) :: f
    integer :: l, k, slot
    n%struct_bytes = int(c_sizeof(n), c_int32_t)
    n%numbnests = numbnests; n%nxmaxn = nxmaxn; n%nymaxn = nymaxn
    n%nxn = 0; n%nyn = 0; n%xln = 0; n%yln = 0; n%xrn = 0; n%yrn = 0; n%xresoln = 1; n%yresoln = 1
    do l = 1, numbnests
      n%nxn(l) = nxn(l); n%nyn(l) = nyn(l)
      n%xln(l) = xln(l); n%yln(l) = yln(l); n%xrn(l) = xrn(l); n%yrn(l) = yrn(l)
      n%xresoln(l) = xresoln(l); n%yresoln(l) = yresoln(l)
    end do
    ierr = fpx_nests_init(flexgpu_handle, n)
    if (ierr /= 0) return
    if (present(geometry_only)) then
      if (geometry_only) return
    end if
    do l = 1, numbnests
      do k = 1, 2
        slot = memind(k)
        f%uu = loc_r(uun(0:,0,1,slot,l)); f%vv = loc_r(vvn(0:,0,1,slot,l)); f%ww = loc_r(wwn(0:,0,1,slot,l))
        f%uupol = c_null_ptr; f%vvpol = c_null_ptr; f%tt = c_null_ptr
        f%rho = loc_r(rhon(0:,0,1,slot,l)); f%drhodz = loc_r(drhodzn(0:,0,1,slot,l))
        f%hmix = loc_r(hmixn(0,0,1,slot,l)); f%ustar = loc_r(ustarn(0,0,1,slot,l)); f%wstar = loc_r(wstarn(0,0,1,slot,l))
        f%oli = loc_r(olin(0,0,1,slot,l)); f%tropopause = loc_r(tropopausen(0,0,1,slot,l))
        f%vdep = loc_r(vdepn(0,0,1,slot,l))
        ierr = fpx_upload_nest_fields(flexgpu_handle, int(l, c_int32_t), int(slot, c_int32_t), f)
        if (ierr /= 0) return
      end do
    end do
  end subroutine flexgpu_upload_nests

  subroutine flexgpu_nests_init(ierr)
    integer, intent(out) :: ierr
    call flexgpu_upload_nests(ierr, geometry_only=.true.)
  end subroutine flexgpu_nests_init

  ! what releaseparticles reads of the nests besides rhon (releaseparticles.f90:216-273): oron and ttn of time slot 2
  subroutine flexgpu_upload_diag_nest_fields(ierr)
    integer, intent(out) :: ierr
    type(fpx_diag_fields) :: f
    integer :: l
    ierr = 0
    do l = 1, numbnests
      f%oro = loc_r(oron(0:,0,l)); f%pv = c_null_ptr; f%qv = c_null_ptr
      f%tt = loc_r(ttn(0:,0,1,2,l))
      ierr = fpx_upload_diag_nest_fields(flexgpu_handle, int(l, c_int32_t), 2_c_int32_t, f)
      if (ierr /= 0) return
    end do
  end subroutine flexgpu_upload_diag_nest_fields

  ! convection inside nested wind fields (convmix.f90:198-250): psn, tt2n, td2n, tthn, qvhn of slot n of every nest, after
  ! readwind_nests filled it and after flexgpu_nests_init / flexgpu_conv_init
  subroutine flexgpu_upload_conv_nest_fields(n, ierr)
    integer, intent(in) :: n
    integer, intent(out) :: ierr
    type(fpx_conv_fields) :: f
    integer :: l
    ierr = 0
    do l = 1, numbnests
      f%ps = loc_r(psn(0,0,1,n,l)); f%tt2 = loc_r(tt2n(0,0,1,n,l)); f%td2 = loc_r(td2n(0,0,1,n,l))
      f%tth = loc_r(tthn(0:,0,1,n,l)); f%qvh = loc_r(qvhn(0:,0,1,n,l))
      f%nuvzmax = nuvzmax; f%reserved = 0
      ierr = fpx_upload_conv_nest_fields(flexgpu_handle, int(l, c_int32_t), int(n, c_int32_t), f)
      if (ierr /= 0) return
    end do
  end subroutine flexgpu_upload_conv_nest_fields

  ! conv_mod cbasefluxn(0:nxmaxn-1,0:nymaxn-1,l) <-> the engine's nest fields
  subroutine flexgpu_cbaseflux_nests(set, ierr)
    logical, intent(in) :: set
    integer, intent(out) :: ierr
    real, allocatable, target :: buf(:,:)
    integer :: l
    ierr = 0
    do l = 1, numbnests
      allocate(buf(0:nxn(l)-1,0:nyn(l)-1))
      if (set) then
        buf = cbasefluxn(0:nxn(l)-1,0:nyn(l)-1,l)
        ierr = fpx_set_cbaseflux_nest(flexgpu_handle, int(l, c_int32_t), c_loc(buf))
      else
        ierr = fpx_get_cbaseflux_nest(flexgpu_handle, int(l, c_int32_t), c_loc(buf))
        if (ierr == 0) cbasefluxn(0:nxn(l)-1,0:nyn(l)-1,l) = buf
      end if
      deallocate(buf)
      if (ierr /= 0) return
    end do
  end subroutine flexgpu_cbaseflux_nests

  ! Replaces `call verttransform_nests(n,uuhn,vvhn,wwhn,pvhn)` (getfields.f90:133,168,184) and the upload of slot n of
  ! every nest; writeback as in flexgpu_verttransform (uun ... drhodzn of slot n).
  subroutine flexgpu_verttransform_nests(n, uuhn, vvhn, wwhn, pvhn, ierr, writeback)
    integer, intent(in) :: n
    real, intent(in) :: uuhn(0:nxmaxn-1,0:nymaxn-1,nuvzmax,maxnests), vvhn(0:nxmaxn-1,0:nymaxn-1,nuvzmax,maxnests)
    real, intent(in) :: pvhn(0:nxmaxn-1,0:nymaxn-1,nuvzmax,maxnests), wwhn(0:nxmaxn-1,0:nymaxn-1,nwzmax,maxnests)
    integer, intent(out) :: ierr
    logical, intent(in), optional :: writeback
    type(fpx_model_levels) :: m
    type(fpx_fields) :: f
    type(fpx_fields_out) :: o
    integer :: l
    logical :: wb
    wb = .true.; if (present(writeback)) wb = writeback
    ierr = 0
    do l = 1, numbnests
      m%uuh = loc_r(uuhn(0:,0,1,l)); m%vvh = loc_r(vvhn(0:,0,1,l)); m%pvh = loc_r(pvhn(0:,0,1,l)); m%wwh = loc_r(wwhn(0:,0,1,l))
      m%tth = loc_r(tthn(0:,0,1,n,l)); m%qvh = loc_r(qvhn(0:,0,1,n,l))
      m%ps = loc_r(psn(0:,0,1,n,l)); m%tt2 = loc_r(tt2n(0:,0,1,n,l)); m%td2 = loc_r(td2n(0:,0,1,n,l))
      m%akz = loc_r(akz); m%bkz = loc_r(bkz); m%aknew = loc_r(aknew); m%bknew = loc_r(bknew)
      m%nuvz = nuvz; m%nwz = nwz; m%init = 0; m%pin_host = 0
      m%nest_dy = dyn(l); m%nest_ylat0 = ylat0n(l)
      f%uu = c_null_ptr; f%vv = c_null_ptr; f%ww = c_null_ptr; f%uupol = c_null_ptr; f%vvpol = c_null_ptr
      f%rho = c_null_ptr; f%drhodz = c_null_ptr; f%tt = c_null_ptr
      f%hmix = loc_r(hmixn(0:,0,1,n,l)); f%ustar = loc_r(ustarn(0:,0,1,n,l)); f%wstar = loc_r(wstarn(0:,0,1,n,l))
      f%oli = loc_r(olin(0:,0,1,n,l)); f%tropopause = loc_r(tropopausen(0:,0,1,n,l))
      f%vdep = loc_r(vdepn(0:,0,1,n,l))
      o%uu = c_null_ptr; o%vv = c_null_ptr; o%ww = c_null_ptr; o%tt = c_null_ptr; o%qv = c_null_ptr
      o%pv = c_null_ptr; o%rho = c_null_ptr; o%drhodz = c_null_ptr; o%uupol = c_null_ptr; o%vvpol = c_null_ptr
      o%height = c_null_ptr; o%nmixz = c_null_ptr
      if (wb) then
        o%uu = loc_r(uun(0:,0,1,n,l)); o%vv = loc_r(vvn(0:,0,1,n,l)); o%ww = loc_r(wwn(0:,0,1,n,l))
        o%tt = loc_r(ttn(0:,0,1,n,l)); o%qv = loc_r(qvn(0:,0,1,n,l)); o%pv = loc_r(pvn(0:,0,1,n,l))
        o%rho = loc_r(rhon(0:,0,1,n,l)); o%drhodz = loc_r(drhodzn(0:,0,1,n,l))
      end if
      ierr = fpx_verttransform_nest(flexgpu_handle, int(l, c_int32_t), int(n, c_int32_t), m, f, o)
      if (ierr /= 0) return
    end do
  end subroutine flexgpu_verttransform_nests
#endif

  subroutine flexgpu_set_windtime(ierr)
    integer, intent(out) :: ierr
    integer(c_int32_t) :: mt(2), mi(2)
    mt = memtime(1:2); mi = memind(1:2)
    ierr = fpx_set_windtime(flexgpu_handle, mt, mi)
  end subroutine flexgpu_set_windtime

  subroutine particle_ptrs(p, j1)
    type(fpx_particles), intent(out) :: p
    integer, intent(in) :: j1
    p%xtra1 = loc_d(xtra1(j1:)); p%ytra1 = loc_d(ytra1(j1:)); p%ztra1 = loc_r(ztra1(j1:))
    p%uap = loc_r(uap(j1:)); p%ucp = loc_r(ucp(j1:)); p%uzp = loc_r(uzp(j1:))
    p%us = loc_r(us(j1:)); p%vs = loc_r(vs(j1:)); p%ws = loc_r(ws(j1:))
    p%itra1 = loc_i(itra1(j1:)); p%itramem = loc_i(itramem(j1:)); p%idt = loc_i(idt(j1:))
    p%npoint = loc_i(npoint(j1:)); p%nclass = loc_i(nclass(j1:)); p%cbt = loc_i2(cbt(j1:))
    p%xmass1 = loc_r(xmass1(j1:,1))
    p%xmass1_ld = size(xmass1, 1)
    p%itrasplit = loc_i(itrasplit(j1:))
    p%xscav_frac1 = c_null_ptr       ! com_mod.f90:683: allocated by readcommand.f90:329,338 for DRYBKDEP / WETBKDEP only
    if (allocated(xscav_frac1)) p%xscav_frac1 = loc_r(xscav_frac1(j1:,1))
  end subroutine particle_ptrs

  ! particles j1..j2 (Fortran numbering) host -> device / device -> host
  ! Replaces `call calcpar(n,uuh,vvh,pvh)` (getfields.f90:128,163,179) for ustar, wstar, oli, hmix, tropopause of slot n
  ! (ECMWF fields): call after flexgpu_verttransform(n, ...) with its sfc argument omitted.  vdep (getvdep) and pv
  ! (calcpv) stay with the host: with DRYDEP the host's vdep(:,:,:,n) must have been computed before this call.
  ! writeback (default .true.): the five fields are also copied into com_mod for host routines that read them.
  ! (No Fortran-host test: calcpar.f90 itself cannot be compiled in the build image -- class_gribfile needs ecCodes.)
  subroutine flexgpu_calcpar(n, ierr, writeback)
    integer, intent(in) :: n
    integer, intent(out) :: ierr
    logical, intent(in), optional :: writeback
    type(fpx_calcpar_in) :: c
    type(fpx_calcpar_out) :: o
    logical :: wb
    wb = .true.; if (present(writeback)) wb = writeback
    c%surfstr = loc_r(surfstr(0,0,1,n)); c%sshf = loc_r(sshf(0,0,1,n))
    c%akm = loc_r(akm); c%bkm = loc_r(bkm)
    c%excessoro = loc_r(excessoro); c%lsubgrid = lsubgrid
    c%vdep = c_null_ptr
    if (DRYDEP) c%vdep = loc_r(vdep(0,0,1,n))
    c%reserved = 0
    o%ustar = c_null_ptr; o%wstar = c_null_ptr; o%oli = c_null_ptr; o%hmix = c_null_ptr; o%tropopause = c_null_ptr
    if (wb) then
      o%ustar = loc_r(ustar(0,0,1,n)); o%wstar = loc_r(wstar(0,0,1,n)); o%oli = loc_r(oli(0,0,1,n))
      o%hmix = loc_r(hmix(0,0,1,n)); o%tropopause = loc_r(tropopause(0,0,1,n))
    end if
    ierr = fpx_calcpar(flexgpu_handle, int(n, c_int32_t), c, o)
  end subroutine flexgpu_calcpar

  ! ---- releaseparticles + the splitting block on the device (SURVEY section 8 f2) ----------------------------
  ! after readreleases (point_mod arrays) and flexgpu_init: hands the release tables to the engine
  subroutine flexgpu_release_init(ierr)
    integer, intent(out) :: ierr
    type(fpx_release) :: r
    r%struct_bytes = int(c_sizeof(r), c_int32_t)
    r%numpoint = numpoint
    r%ireleasestart = loc_i(ireleasestart); r%ireleaseend = loc_i(ireleaseend); r%kindz = loc_i2(kindz)
    r%xpoint1 = loc_r(xpoint1); r%xpoint2 = loc_r(xpoint2); r%ypoint1 = loc_r(ypoint1); r%ypoint2 = loc_r(ypoint2)
    r%zpoint1 = loc_r(zpoint1); r%zpoint2 = loc_r(zpoint2)
    r%point_hour = loc_r(point_hour); r%area_hour = loc_r(area_hour); r%point_dow = loc_r(point_dow); r%area_dow = loc_r(area_dow)
    r%bdate = bdate
    r%itsplit = itsplit; r%ind_rel = ind_rel; r%nclassunc = nclassunc
    r%reserved = 0
    ierr = fpx_release_init(flexgpu_handle, r)
  end subroutine flexgpu_release_init

  ! replaces `call releaseparticles(itime)` (timemanager.f90:246): numpart, numparticlecount (com_mod), xmasssave
  ! (xmass_mod) and rho_rel (point_mod) are updated in place, the new particles exist on the device only
  subroutine flexgpu_releaseparticles(itime, ierr)
    integer, intent(in) :: itime
    integer, intent(out) :: ierr
    integer(c_int64_t) :: np
    integer(c_int32_t) :: npc
    np = numpart; npc = numparticlecount
    ierr = fpx_releaseparticles(flexgpu_handle, int(itime, c_int32_t), np, npc, loc_r(xmasssave), loc_r(rho_rel), c_null_ptr)
    if (ierr /= 0) return
    numpart = int(np); numparticlecount = npc
  end subroutine flexgpu_releaseparticles

  ! replaces the splitting block of the time manager (timemanager.f90:473-504); itrasplit lives on the device
  subroutine flexgpu_split_particles(itime, ierr)
    integer, intent(in) :: itime
    integer, intent(out) :: ierr
    integer(c_int64_t) :: np
    np = numpart
    ierr = fpx_split_particles(flexgpu_handle, int(itime, c_int32_t), np)
    if (ierr /= 0) return
    numpart = int(np)
  end subroutine flexgpu_split_particles

  ! ---- mpi_mod.f90:566-856: levelling the processes' particle counts; the host keeps its MPI calls ----
  ! mpif_calculate_part_redist without the MPI_Allgather: npart_per_process(0:np-1) -> what THIS process does
  ! (role 0 nothing, 1 send, 2 receive num_trans particles to / from process peer)
  subroutine flexgpu_redist_plan(npart_per_process, mp_np, mp_partid, role, peer, num_trans, ierr)
    integer, intent(in) :: npart_per_process(0:), mp_np, mp_partid
    integer, intent(out) :: role, peer, num_trans, ierr
    integer(c_int64_t) :: cnt(0:mp_np-1), nt
    integer(c_int32_t) :: r, p
    cnt = int(npart_per_process(0:mp_np-1), c_int64_t)
    ierr = fpx_redist_plan(cnt, int(mp_np, c_int32_t), int(mp_partid, c_int32_t), int(ipout, c_int32_t), r, p, nt)
    role = int(r); peer = int(p); num_trans = int(nt)
  end subroutine flexgpu_redist_plan

  ! bytes of the one message of num_trans particles: 64-bit (56 B per particle and more: a default integer overflows from
  ! about 3.8e7 particles on)
  integer(c_int64_t) function flexgpu_redist_bytes(num_trans)
    integer, intent(in) :: num_trans
    flexgpu_redist_bytes = fpx_redist_bytes(flexgpu_handle, int(num_trans, c_int64_t))
  end function flexgpu_redist_bytes

  ! the sending half of mpif_redist_part (:700-746): buf(1:flexgpu_redist_bytes(num_trans)) is the ONE message
  subroutine flexgpu_redist_pack(itime, num_trans, buf, ierr)
    integer, intent(in) :: itime, num_trans
    integer(c_int8_t), intent(inout), target :: buf(:)
    integer, intent(out) :: ierr
    integer(c_int64_t) :: np
    np = numpart
    ierr = fpx_redist_pack(flexgpu_handle, int(itime, c_int32_t), int(num_trans, c_int64_t), c_loc(buf), int(size(buf), c_int64_t), np)
    if (ierr == 0) numpart = int(np)
  end subroutine flexgpu_redist_pack

  ! the receiving half (:749-841)
  subroutine flexgpu_redist_unpack(itime, num_trans, buf, ierr)
    integer, intent(in) :: itime, num_trans
    integer(c_int8_t), intent(in), target :: buf(:)
    integer, intent(out) :: ierr
    integer(c_int64_t) :: np
    np = numpart
    ierr = fpx_redist_unpack(flexgpu_handle, int(itime, c_int32_t), int(num_trans, c_int64_t), c_loc(buf), int(size(buf), c_int64_t), np)
    if (ierr == 0) numpart = int(np)
  end subroutine flexgpu_redist_unpack

  subroutine flexgpu_upload_particles(j1, j2, ierr)
    integer, intent(in) :: j1, j2
    integer, intent(out) :: ierr
    type(fpx_particles) :: p
    call particle_ptrs(p, j1)
    ierr = fpx_upload_particles(flexgpu_handle, int(j1-1, c_int64_t), int(j2-j1+1, c_int64_t), p)
  end subroutine flexgpu_upload_particles

  subroutine flexgpu_download_particles(j1, j2, ierr)
    integer, intent(in) :: j1, j2
    integer, intent(out) :: ierr
    type(fpx_particles) :: p
    call particle_ptrs(p, j1)
    ierr = fpx_download_particles(flexgpu_handle, int(j1-1, c_int64_t), int(j2-j1+1, c_int64_t), p)
  end subroutine flexgpu_download_particles

  ! the replacement of the particle loop timemanager.f90:531-712
  subroutine flexgpu_step(itime, stats, ierr)
    integer, intent(in) :: itime
    type(fpx_step_stats), intent(out) :: stats
    integer, intent(out) :: ierr
    ierr = fpx_step(flexgpu_handle, int(itime, c_int32_t), stats)
    nan_count = nan_count + int(stats%nan_count)      ! com_mod counters, advance.f90:421,439
    nan_count2 = nan_count2 + int(stats%nan_count2)
  end subroutine flexgpu_step

  ! ---- output grids (rows a21-a23 of the hot path): OUTGRID / OUTGRID_NEST / RECEPTORS state of com_mod ----
  function loc_i1(x) result(p)
    integer(kind=1), target, intent(in) :: x(*)
    type(c_ptr) :: p
    p = c_loc(x)
  end function loc_i1
  function loc_dep(x) result(p)
    real(dep_prec), target, intent(in) :: x(*)
    type(c_ptr) :: p
    p = c_loc(x)
  end function loc_dep

  ! after readoutgrid/outgrid_init (and their _nest twins, readreceptors): device copies of gridunc, drygridunc,
  ! wetgridunc (+ griduncn ... when nested_output = 1, creceptor when numreceptor > 0)
  subroutine flexgpu_outgrid_init(loutnext, ierr)
    integer, intent(in) :: loutnext
    integer, intent(out) :: ierr
    type(fpx_outgrid) :: g
    type(fpx_outgrid_nest) :: gn
    g%struct_bytes = int(c_sizeof(g), c_int32_t)
    g%numxgrid = numxgrid; g%numygrid = numygrid; g%numzgrid = numzgrid
    g%dxout = dxout; g%dyout = dyout; g%xoutshift = xoutshift; g%youtshift = youtshift
    g%maxpointspec_act = maxpointspec_act; g%nclassunc = nclassunc; g%nageclass = nageclass
    g%lage = 0; g%lage(1:nageclass) = lage(1:nageclass)
    g%ind_samp = ind_samp; g%ioutputforeachrelease = ioutputforeachrelease
    g%lusekerneloutput = merge(1, 0, lusekerneloutput)
    g%reserved = 0
    ierr = fpx_outgrid_init(flexgpu_handle, g, loc_r(outheight))
    if (ierr /= 0) return
    ierr = fpx_set_output_times(flexgpu_handle, int(loutnext, c_int32_t), int(loutstep, c_int32_t))
    if (ierr /= 0) return
    if (nested_output .eq. 1) then
      gn%struct_bytes = int(c_sizeof(gn), c_int32_t)
      gn%numxgridn = numxgridn; gn%numygridn = numygridn
      gn%dxoutn = dxoutn; gn%dyoutn = dyoutn; gn%xoutshiftn = xoutshiftn; gn%youtshiftn = youtshiftn
      gn%reserved = 0
      ierr = fpx_outgrid_nest_init(flexgpu_handle, gn)
      if (ierr /= 0) return
    end if
    if (numreceptor .gt. 0) then
      ierr = fpx_receptors_init(flexgpu_handle, int(numreceptor, c_int32_t), loc_r(xreceptor), loc_r(yreceptor), &
                                loc_r(receptorarea))
    end if
  end subroutine flexgpu_outgrid_init

  ! replaces "call conccalc(itime,weight)" (timemanager.f90:350-365); particles stay on the device
  subroutine flexgpu_conccalc(itime, weight, ierr)
    integer, intent(in) :: itime
    real, intent(in) :: weight
    integer, intent(out) :: ierr
    ierr = fpx_conccalc(flexgpu_handle, int(itime, c_int32_t), real(weight, c_double))
  end subroutine flexgpu_conccalc

  ! Before the host's own concoutput: the device's sums overwrite the host's gridunc, drygridunc, wetgridunc
  ! (+ the nested grids and creceptor).  clear = 1 zeroes the device's gridunc / griduncn / creceptor, as
  ! concoutput.f90:719-720 zeroes the host's after writing; the deposition grids keep accumulating on the device
  ! exactly as drygridunc / wetgridunc do on the host (zeroed once, outgrid_init.f90:317-318), so every call hands
  ! the host the cumulative values its concoutput expects.  allreduce = .true. (several ranks, after
  ! flexgpu_comm_init_host or fpx_comm_init): the arrays receive the sums over all ranks (what
  ! mpif_tm_reduce_grid leaves in gridunc0 ... on the root, mpi_mod.f90:2451-2492); the ranks' partial sums stay
  ! on the device.
  subroutine flexgpu_get_grids(clear, ierr, allreduce)
    integer, intent(in) :: clear
    integer, intent(out) :: ierr
    logical, intent(in), optional :: allreduce
    integer(c_int32_t) :: ar
    ar = 0
    if (present(allreduce)) ar = merge(1, 0, allreduce)
    ierr = fpx_get_grids(flexgpu_handle, loc_r(gridunc), loc_dep(drygridunc), ar, int(clear, c_int32_t))
    if (ierr /= 0) return
    ierr = fpx_get_wetgrid(flexgpu_handle, loc_dep(wetgridunc), ar)
    if (ierr /= 0) return
    if (nested_output .eq. 1) then
      ierr = fpx_get_grids_nest(flexgpu_handle, loc_r(griduncn), loc_dep(drygriduncn), loc_dep(wetgriduncn), &
                                ar, int(clear, c_int32_t))
      if (ierr /= 0) return
    end if
    if (numreceptor .gt. 0) then
      ierr = fpx_get_receptors(flexgpu_handle, loc_r(creceptor), int(maxreceptor, c_int32_t), ar, &
                               int(clear, c_int32_t))
    end if
  end subroutine flexgpu_get_grids

  ! The grid reduction over the ranks of an MPI host through the host's own all-reduce (the transport
  ! mpi_mod.f90:2471-2492 uses): fn = c_funloc of a bind(C) function
  !   integer(c_int) function f(user, send, recv, count, dtype)   ! dtype 0: 4-byte reals, 1: 8-byte reals
  ! that calls MPI_Allreduce(send, recv, count, MPI_REAL4|MPI_REAL8, MPI_SUM, comm) and returns 0.
  subroutine flexgpu_comm_init_host(nranks, rank, fn, ierr)
    integer, intent(in) :: nranks, rank
    type(c_funptr), intent(in) :: fn
    integer, intent(out) :: ierr
    ierr = fpx_comm_init_host(flexgpu_handle, int(nranks, c_int32_t), int(rank, c_int32_t), fn, c_null_ptr)
  end subroutine flexgpu_comm_init_host

  ! numpart_tot_mpi of timemanager_mpi.f90:552-562: numpart summed over the ranks (and, next to it, the particles that
  ! are still alive); allreduce = .false. returns this rank's own counts
  subroutine flexgpu_count_particles(allreduce, nlive, numpart_tot, ierr)
    logical, intent(in) :: allreduce
    integer(c_int64_t), intent(out) :: nlive, numpart_tot
    integer, intent(out) :: ierr
    integer(c_int64_t) :: loc(2), tot(2)
    ierr = fpx_count_particles(flexgpu_handle, loc, tot, merge(1_c_int32_t, 0_c_int32_t, allreduce))
    nlive = tot(1); numpart_tot = tot(2)
  end subroutine flexgpu_count_particles

  ! a tuning / diagnostic knob of the engine (include/flexpart_amd.h: fpx_set_option; none changes a result), and what the
  ! engine decided ("time_blended_packs", "blended_steps", "pbl_launches_per_step")
  subroutine flexgpu_set_option(name, value, ierr)
    character(len=*), intent(in) :: name, value
    integer, intent(out) :: ierr
    ierr = fpx_set_option(flexgpu_handle, trim(name)//c_null_char, trim(value)//c_null_char)
  end subroutine flexgpu_set_option

  subroutine flexgpu_get_info(name, value, ierr)
    character(len=*), intent(in) :: name
    integer(c_int64_t), intent(out) :: value
    integer, intent(out) :: ierr
    ierr = fpx_get_info(flexgpu_handle, trim(name)//c_null_char, value)
  end subroutine flexgpu_get_info

  ! ---- wet deposition: species parameters of readspecies.f90, fields of readwind/verttransform ----
  subroutine flexgpu_wet_init(ierr)
    integer, intent(out) :: ierr
    type(fpx_wet_config) :: w
    integer :: ks
    w%struct_bytes = int(c_sizeof(w), c_int32_t)
    w%wetdepspec = 0; w%weta_gas = 0; w%wetb_gas = 0; w%crain_aero = 0; w%csnow_aero = 0
    w%ccn_aero = 0; w%in_aero = 0; w%henry = 0
    do ks = 1, nspec
      w%wetdepspec(ks) = merge(1, 0, WETDEPSPEC(ks))
      w%weta_gas(ks) = weta_gas(ks); w%wetb_gas(ks) = wetb_gas(ks)
      w%crain_aero(ks) = crain_aero(ks); w%csnow_aero(ks) = csnow_aero(ks)
      w%ccn_aero(ks) = ccn_aero(ks); w%in_aero(ks) = in_aero(ks); w%henry(ks) = henry(ks)
    end do
    w%readclouds = merge(1, 0, readclouds)
    w%reserved = 0
    ierr = fpx_wet_init(flexgpu_handle, w)
  end subroutine flexgpu_wet_init

  subroutine flexgpu_upload_wet_fields(slot, ierr)
    integer, intent(in) :: slot
    integer, intent(out) :: ierr
    type(fpx_wet_fields) :: f
    f%lsprec = loc_r(lsprec(0,0,1,slot)); f%convprec = loc_r(convprec(0,0,1,slot)); f%tcc = loc_r(tcc(0,0,1,slot))
    f%ctwc = c_null_ptr
    if (readclouds) f%ctwc = loc_r(ctwc(0,0,slot))
    f%tt = loc_r(tt(0,0,1,slot))
    f%clouds = loc_i1(clouds(0,0,1,slot)); f%cloudsh = loc_i(cloudsh(0,0,slot))
    ierr = fpx_upload_wet_fields(flexgpu_handle, int(slot, c_int32_t), f)
  end subroutine flexgpu_upload_wet_fields

#ifdef FLEXGPU_NESTS
  subroutine flexgpu_upload_wet_nest_fields(slot, ierr)
    integer, intent(in) :: slot
    integer, intent(out) :: ierr
    type(fpx_wet_fields) :: f
    integer :: l
    ierr = 0
    do l = 1, numbnests
      f%lsprec = loc_r(lsprecn(0,0,1,slot,l)); f%convprec = loc_r(convprecn(0,0,1,slot,l)); f%tcc = loc_r(tccn(0,0,1,slot,l))
      f%ctwc = c_null_ptr
      if (readclouds_nest(l)) f%ctwc = loc_r(ctwcn(0:,0,slot,l))
      f%tt = loc_r(ttn(0:,0,1,slot,l))
      f%clouds = loc_i1(cloudsn(0:,0,1,slot,l)); f%cloudsh = loc_i(cloudshn(0:,0,slot,l))
      ierr = fpx_upload_wet_nest_fields(flexgpu_handle, int(l, c_int32_t), int(slot, c_int32_t), f, &
                                        int(merge(1, 0, readclouds_nest(l)), c_int32_t))
      if (ierr /= 0) return
    end do
  end subroutine flexgpu_upload_wet_nest_fields
#endif

  ! replaces "call wetdepo(itime,lsynctime,loutnext)" (timemanager.f90:164-169)
  subroutine flexgpu_wetdepo(itime, ltsample, loutnext, ierr)
    integer, intent(in) :: itime, ltsample, loutnext
    integer, intent(out) :: ierr
    ierr = fpx_wetdepo(flexgpu_handle, int(itime, c_int32_t), int(ltsample, c_int32_t), int(loutnext, c_int32_t))
  end subroutine flexgpu_wetdepo

end module flexgpu_mod
