! TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
!
! ref_conv_driver: runs the *unmodified* reference routines of the convective mixing that compile in this image --
! CONVECT / TLIFT (convect43c.f90), redist (redist.f90), sort2 (sort2.f90), f_qvsat (qvsat.f90), ew (ew.f90), ran3
! (random_mod.f90), with par_mod / com_mod / conv_mod, all compiled where they lie by oracle/build_ref.sh -- over a
! scenario file, so that oracle/convect_oracle.c can be pinned against them (SURVEY section 8 f3).
! convmix.f90 and calcmatrix.f90 themselves `use class_gribfile` (ecCodes' grib_api) and cannot be built here; the two
! subroutines below, glue_calcmatrix and the particle loop of the main program, are OUR restatement of their ECMWF branch
! (calcmatrix.f90:56-137, convmix.f90:61-196) around the real CONVECT and REDIST.  This file is our own code: it
! contains no reference source.
!
! Usage:  convref_rK in.bin out.bin [gpu]
!   gpu: the same com_mod / conv_mod arrays go to the MI355X engine through flexpart_amd/fortran/flexgpu_mod.f90
!        (flexgpu_conv_init / flexgpu_upload_conv_fields / flexgpu_convmix replace the loop below), serial-stream parity mode.
program convref
  use par_mod
  use com_mod
  use conv_mod
  use flexgpu_mod
  implicit none
  character(len=512) :: fin, fout, arg3, gmsg
  integer :: use_gpu, gerr
  integer(kind=4) :: nest_on, nxnl, nynl
  real(kind=8) :: ngeom(6)
  real(kind=8), allocatable :: psn8(:,:,:), tt2n8(:,:,:), td2n8(:,:,:), tthn8(:,:,:,:), qvhn8(:,:,:,:), cbn8(:,:)
  integer, allocatable :: igridn(:)
  real, allocatable :: cbln(:,:)
  real :: xtn, ytn
  real, parameter :: epsn = nxmax / 3.e5
  integer :: ngrid, pass, gnx
  integer(kind=4) :: hdr(12)
  integer :: nxl, nyl, nuvzl, ncalls, fmcap, n, ic, i, j, k, kk
  real(kind=8) :: hnz
  real(kind=8), allocatable :: akz8(:), bkz8(:), akm8(:), bkm8(:), ps8(:,:,:), tt28(:,:,:), td28(:,:,:)
  real(kind=8), allocatable :: tth8(:,:,:,:), qvh8(:,:,:,:), cb8(:,:), x8(:), y8(:), z8(:), fm8(:,:,:)
  integer(kind=4), allocatable :: itimes(:), due(:,:), lconvcol(:,:), ntopcol(:,:), fmcol(:)
  integer, allocatable :: igrid(:), ipoint(:)
  real, allocatable :: cbl(:,:)
  integer(kind=4) :: fmcount
  integer :: itime, igr, igrold, ipart, kpart, ix, jy, ktop, ipconv, kz
  logical :: lconv
  real :: x, y, dt1, dt2, dtt, delt
  integer :: mt1, mt2

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  use_gpu = 0
  if (command_argument_count() >= 3) then
    call get_command_argument(3, arg3)
    if (trim(arg3) == 'gpu') use_gpu = 1
  end if
  open(31, file=trim(fin), access='stream', form='unformatted', status='old')
  read(31) hdr
  nxl = hdr(1); nyl = hdr(2); nuvzl = hdr(3); nconvlev = hdr(4); ldirect = hdr(5); lsynctime = hdr(6)
  mt1 = hdr(7); mt2 = hdr(8); n = hdr(9); ncalls = hdr(10); fmcap = hdr(11); nz = hdr(12)
  if (nuvzl > nuvzmax .or. nconvlev > nconvlevmax - 1 .or. n > maxpart .or. nz > nzmax) then
    print *, 'convref: scenario larger than the compile-time sizes of par_mod'
    stop 2
  end if
  read(31) hnz
  allocate(akz8(nuvzl), bkz8(nuvzl), akm8(nuvzl), bkm8(nuvzl), ps8(nxl,nyl,2), tt28(nxl,nyl,2), td28(nxl,nyl,2))
  allocate(tth8(nxl,nyl,nuvzl,2), qvh8(nxl,nyl,nuvzl,2), cb8(nxl,nyl), x8(n), y8(n), z8(n))
  allocate(itimes(ncalls), due(n,ncalls), lconvcol(nxl,nyl), ntopcol(nxl,nyl), fmcol(fmcap), fm8(nconvlev,nconvlev,fmcap))
  allocate(igrid(n), ipoint(n), cbl(nxl,nyl))
  read(31) akz8, bkz8, akm8, bkm8, ps8, tt28, td28, tth8, qvh8, cb8, x8, y8, z8, itimes, due
  nest_on = 0
  read(31, iostat=i) nest_on
  if (i /= 0) nest_on = 0
  allocate(igridn(n))
  if (nest_on == 1) then
    read(31) nxnl, nynl
    read(31) ngeom
    allocate(psn8(nxnl,nynl,2), tt2n8(nxnl,nynl,2), td2n8(nxnl,nynl,2), tthn8(nxnl,nynl,nuvzl,2), qvhn8(nxnl,nynl,nuvzl,2), cbn8(nxnl,nynl), cbln(nxnl,nynl))
    read(31) psn8, tt2n8, td2n8, tthn8, qvhn8, cbn8
    cbln = real(cbn8)
#ifndef FLEXREF_NESTS
    print *, 'convref: this build has no nests (par_mod maxnests = 0)'
    stop 2
#endif
  end if
  close(31)
  nuvz = nuvzl
  do k = 1, nuvzl
    akz(k) = real(akz8(k)); bkz(k) = real(bkz8(k)); akm(k) = real(akm8(k)); bkm(k) = real(bkm8(k))
  end do
  height(nz) = real(hnz)
  numpart = n
  call com_mod_allocate_part(n)
  do i = 1, n
    xtra1(i) = x8(i); ytra1(i) = y8(i); ztra1(i) = real(z8(i))
  end do
  cbl = real(cb8)

  open(32, file=trim(fout), access='stream', form='unformatted', status='replace')
  if (use_gpu == 1) then
    ! what flexgpu_init reads of com_mod (no trajectory step is taken): a limited-area grid of nxl x nyl cells, nz levels
    nx = nxl; ny = nyl; nxmin1 = nx - 1; nymin1 = ny - 1
    dx = 1.; dy = 1.; xlon0 = -20.; ylat0 = 20.; xglobal = .false.; nglobal = .false.; sglobal = .false.
    switchnorthg = 999999.; switchsouthg = 999999.
    ipout = 0; iflux = 0; linit_cond = 0
    do k = 1, nz
      height(k) = real(hnz) * real(k - 1) / real(nz - 1)
    end do
    nmixz = nz
    method = 1; ctl = 0.2; ifine = 4; turbswitch = .true.; cblflag = 0; mintime = 1
    mdomainfill = 0; lsettling = .false.; DRYDEP = .false.; nageclass = 1; lage(1) = 999999999; nspec = 1; mquasilag = 0
    memtime(1) = mt1; memtime(2) = mt2; memind(1) = 1; memind(2) = 2; lwindinterv = abs(mt2 - mt1)
    do j = 1, nyl
      do i = 1, nxl
        do k = 1, 2
          ps(i-1,j-1,1,k) = real(ps8(i,j,k)); tt2(i-1,j-1,1,k) = real(tt28(i,j,k)); td2(i-1,j-1,1,k) = real(td28(i,j,k))
          tth(i-1,j-1,1:nuvzl,k) = real(tth8(i,j,1:nuvzl,k)); qvh(i-1,j-1,1:nuvzl,k) = real(qvh8(i,j,1:nuvzl,k))
        end do
      end do
    end do
    itra1(:) = -999999999; itramem(:) = 0; idt(:) = 1; npoint(:) = 1; nclass(:) = 1; itrasplit(:) = 999999999
    uap(:) = 0.; ucp(:) = 0.; uzp(:) = 0.; us(:) = 0.; vs(:) = 0.; ws(:) = 0.; cbt(:) = 1; xmass1(:,:) = 1.
    call flexgpu_init(gerr, nmaxpart=n)
    if (gerr == 0) call flexgpu_use_table_rng(gerr)
    if (gerr == 0) call flexgpu_set_windtime(gerr)
    if (gerr == 0) call flexgpu_conv_init(gerr)
    if (gerr == 0) call flexgpu_upload_conv_fields(1, gerr)
    if (gerr == 0) call flexgpu_upload_conv_fields(2, gerr)
    if (gerr /= 0) call gpu_fail('set-up')
    cbaseflux(0:nxl-1,0:nyl-1) = cbl
    call flexgpu_cbaseflux(.true., gerr)
    if (gerr /= 0) call gpu_fail('flexgpu_cbaseflux')
  end if
  do ic = 1, ncalls
    itime = itimes(ic)
    if (use_gpu == 1) then
      do i = 1, n
        itra1(i) = merge(itime, itime + 12345, due(i, ic) /= 0)
      end do
      call flexgpu_upload_particles(1, n, gerr)
      if (gerr == 0) call flexgpu_convmix(itime, gerr)
      if (gerr == 0) call flexgpu_download_particles(1, n, gerr)
      if (gerr == 0) call flexgpu_cbaseflux(.false., gerr)
      if (gerr /= 0) call gpu_fail('flexgpu_convmix')
      do i = 1, n
        z8(i) = ztra1(i)
      end do
      cb8 = cbaseflux(0:nxl-1,0:nyl-1)
      lconvcol = -1; ntopcol = 0; fmcount = 0; fm8 = 0.d0; fmcol = -1
      write(32) z8, cb8, lconvcol, ntopcol, fmcount, fmcol, fm8
      cycle
    end if
    dt1 = real(itime - mt1)
    dt2 = real(mt2 - itime)
    dtt = 1. / (dt1 + dt2)
    delt = real(abs(lsynctime))
    lconvcol = -1; ntopcol = 0; fmcount = 0; fm8 = 0.d0; fmcol = -1
    lconv = .false.
    ! convmix.f90:92-135 (mother grid)
    do ipart = 1, n
      igrid(ipart) = -1
      igridn(ipart) = -1
      ipoint(ipart) = ipart
      if (due(ipart, ic) == 0) cycle
      x = xtra1(ipart)
      y = ytra1(ipart)
      ngrid = 0
      ! convmix.f90:104-111 (ECMWF input: with eps)
      if (nest_on == 1) then
        if (x > real(ngeom(1)) + epsn .and. x < real(ngeom(3)) - epsn .and. y > real(ngeom(2)) + epsn .and. y < real(ngeom(4)) - epsn) ngrid = 1
      end if
      if (ngrid > 0) then
        xtn = (x - real(ngeom(1))) * real(ngeom(5))
        ytn = (y - real(ngeom(2))) * real(ngeom(6))
        ix = nint(xtn)
        jy = nint(ytn)
        igridn(ipart) = 1 + jy * nxnl + ix
      else
        ix = nint(x)
        jy = nint(y)
        igrid(ipart) = 1 + jy * nxl + ix
      end if
    end do
    do pass = 0, nest_on
    gnx = nxl
    if (pass == 1) then      ! convmix.f90:198-203
      gnx = nxnl
      do ipart = 1, n
        ipoint(ipart) = ipart
        igrid(ipart) = igridn(ipart)
      end do
    end if
    call sort2(n, igrid, ipoint)
    igrold = -1
    ktop = 0
    do kpart = 1, n
      igr = igrid(kpart)
      if (igr == -1) cycle
      ipart = ipoint(kpart)
      if (igr /= igrold .and. pass == 1) then
        jy = (igr - 1) / gnx
        ix = igr - jy * gnx - 1
        ! convmix.f90:214-228: the nest's profiles
        psconv = (real(psn8(ix+1,jy+1,1)) * dt2 + real(psn8(ix+1,jy+1,2)) * dt1) * dtt
        tt2conv = (real(tt2n8(ix+1,jy+1,1)) * dt2 + real(tt2n8(ix+1,jy+1,2)) * dt1) * dtt
        td2conv = (real(td2n8(ix+1,jy+1,1)) * dt2 + real(td2n8(ix+1,jy+1,2)) * dt1) * dtt
        do kz = 1, nuvzl - 1
          tconv(kz) = (real(tthn8(ix+1,jy+1,kz+1,1)) * dt2 + real(tthn8(ix+1,jy+1,kz+1,2)) * dt1) * dtt
          qconv(kz) = (real(qvhn8(ix+1,jy+1,kz+1,1)) * dt2 + real(qvhn8(ix+1,jy+1,kz+1,2)) * dt1) * dtt
        end do
        call glue_calcmatrix(lconv, delt, cbln(ix+1,jy+1))
        igrold = igr
        ktop = 0
      else if (igr /= igrold) then
        jy = (igr - 1) / nxl
        ix = igr - jy * nxl - 1
        ! convmix.f90:154-166: the column's profiles at the particle time
        psconv = (real(ps8(ix+1,jy+1,1)) * dt2 + real(ps8(ix+1,jy+1,2)) * dt1) * dtt
        tt2conv = (real(tt28(ix+1,jy+1,1)) * dt2 + real(tt28(ix+1,jy+1,2)) * dt1) * dtt
        td2conv = (real(td28(ix+1,jy+1,1)) * dt2 + real(td28(ix+1,jy+1,2)) * dt1) * dtt
        do kz = 1, nuvzl - 1
          tconv(kz) = (real(tth8(ix+1,jy+1,kz+1,1)) * dt2 + real(tth8(ix+1,jy+1,kz+1,2)) * dt1) * dtt
          qconv(kz) = (real(qvh8(ix+1,jy+1,kz+1,1)) * dt2 + real(qvh8(ix+1,jy+1,kz+1,2)) * dt1) * dtt
        end do
        call glue_calcmatrix(lconv, delt, cbl(ix+1,jy+1))
        lconvcol(ix+1,jy+1) = merge(1, 0, lconv)
        if (lconv) ntopcol(ix+1,jy+1) = nconvtop
        if (lconv .and. fmcount < fmcap) then
          fmcount = fmcount + 1
          fmcol(fmcount) = jy * nxl + ix
          do kk = 1, nconvlev
            do k = 1, nconvlev
              if (k <= nconvtop .and. kk <= nconvtop) fm8(k, kk, fmcount) = fmassfrac(k, kk)
            end do
          end do
        end if
        igrold = igr
        ktop = 0
      end if
      if (lconv) then
        itra1(ipart) = itime
        call redist(ipart, ktop, ipconv)
      end if
    end do
    end do     ! pass
    do i = 1, n
      z8(i) = ztra1(i)
    end do
    cb8 = cbl
    write(32) z8, cb8, lconvcol, ntopcol, fmcount, fmcol, fm8
    if (nest_on == 1) then
      cbn8 = cbln
      write(32) cbn8
    end if
  end do
  close(32)
  if (use_gpu == 1) call flexgpu_finalize()

contains

  subroutine gpu_fail(where)
    character(len=*), intent(in) :: where
    call flexgpu_last_error(gmsg)
    write(*,*) 'convref gpu: ', where, ': ', trim(gmsg)
    stop 3
  end subroutine gpu_fail

  ! our restatement of calcmatrix.f90:56-137 (ECMWF branch) around the reference's CONVECT
  subroutine glue_calcmatrix(lconv, delt, cbmf)
    logical, intent(out) :: lconv
    real, intent(in) :: delt
    real, intent(inout) :: cbmf
    real :: rlevmass, summe, cbmfold, precip, qprime, tprime, wd, f_qvsat
    integer :: iflag, k, kk, kuvz
    lconv = .false.
    phconv(1) = psconv
    do kuvz = 2, nuvz
      k = kuvz - 1
      pconv(k) = (akz(kuvz) + bkz(kuvz) * psconv)
      phconv(kuvz) = (akm(kuvz) + bkm(kuvz) * psconv)
      dpr(k) = phconv(k) - phconv(kuvz)
      qsconv(k) = f_qvsat(pconv(k), tconv(k))
      do kk = 1, nconvlev
        fmassfrac(k, kk) = 0.
      end do
    end do
    cbmfold = cbmf
    do k = 1, nconvlev + 1
      pconv_hpa(k) = pconv(k) / 100.
      phconv_hpa(k) = phconv(k) / 100.
    end do
    phconv_hpa(nconvlev + 1) = phconv(nconvlev + 1) / 100.
    call convect(nconvlevmax, nconvlev, delt, iflag, precip, wd, tprime, qprime, cbmf)
    if (iflag /= 1 .and. iflag /= 4) then
      cbmf = cbmfold
      return
    end if
    if (cbmf <= 0. .and. cbmfold <= 0.) then
      cbmf = cbmfold
      return
    end if
    lconv = .true.
    do k = 1, nconvtop
      rlevmass = dpr(k) / ga
      summe = 0.
      do kk = 1, nconvtop
        fmassfrac(k, kk) = delt * fmass(k, kk)
        summe = summe + fmassfrac(k, kk)
      end do
      fmassfrac(k, k) = fmassfrac(k, k) + rlevmass - summe
    end do
  end subroutine glue_calcmatrix

end program convref
