! TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
!
! ref_vt_driver: drives the *unmodified* reference routine verttransform_ecmwf
! (/root/reference/src/verttransform_ecmwf.f90, compiled where it lies by
! oracle/build_ref.sh) on model-level input read from a scenario file and dumps the
! z-level fields it produces (SURVEY section 8 f1: eta -> z, rho, drhodz, polar winds).
! This file is our own code: it contains no reference source, only calls into it and
! assignments to its module variables.
!
! Usage:  vtref_rK scenario.bin out.bin [gpu]
!   gpu: the same com_mod arrays go through the MI355X engine (flexgpu_verttransform of
!        flexpart_amd/fortran/flexgpu_mod.f90, ISO_C_BINDING) instead of the Fortran routine --
!        the drop-in integration test (needs a GPU; run on the GPU box).
! Record format as oracle/ref_driver.f90: {name*16, dtype i4 (1=i32, 2=f64), count i8, payload}.
! 3-D input arrays travel compact, (nx,ny,nlev) x fastest, as f64.

module vt_io
  implicit none
  integer, parameter :: uin=31, uout=32
contains
  subroutine put_i(name, a, n)
    character(len=*), intent(in) :: name
    integer, intent(in) :: n
    integer, intent(in) :: a(n)
    character(len=16) :: nm
    nm = name
    write(uout) nm, 1_4, int(n,8), a(1:n)
  end subroutine put_i
  subroutine put_d(name, a, n)
    character(len=*), intent(in) :: name
    integer, intent(in) :: n
    real(kind=8), intent(in) :: a(n)
    character(len=16) :: nm
    nm = name
    write(uout) nm, 2_4, int(n,8), a(1:n)
  end subroutine put_d
end module vt_io

program vtref
  use par_mod
  use com_mod
  use cmapf_mod
  use point_mod
  use vt_io
  use flexgpu_mod
  implicit none

  integer :: use_gpu, gerr, abi_sizes(10)
  character(len=256) :: gmsg
  character(len=512) :: arg3
  character(len=512) :: fscen, fout
  character(len=16) :: name
  integer(kind=4) :: dtype
  integer(kind=8) :: cnt
  integer, allocatable :: ibuf(:)
  real(kind=8), allocatable :: dbuf(:), tmp(:)
  real, allocatable :: uuh(:,:,:), vvh(:,:,:), pvh(:,:,:), wwh(:,:,:)
#ifdef FLEXREF_NESTS
  real, allocatable :: uuhn(:,:,:,:), vvhn(:,:,:,:), pvhn(:,:,:,:), wwhn(:,:,:,:)
  integer :: gnxn, gnyn
#endif
  integer :: ios, n, gnx, gny, gnz, ncalls, icall
  real :: sizenorth, sizesouth
  integer(kind=8) :: c0, c1, crate

  call get_command_argument(1, fscen)
  if (trim(fscen) .eq. 'abi') then      ! vtref_rK abi: sizes of the Fortran bind(C) types, for the ABI test
    call flexgpu_abi_sizes(abi_sizes)
    write(*,'(10i6)') abi_sizes
    stop
  end if
  call get_command_argument(2, fout)
  use_gpu = 0
  if (command_argument_count() .ge. 3) then
    call get_command_argument(3, arg3)
    if (trim(arg3) .eq. 'gpu') use_gpu = 1
  end if

  allocate(uuh(0:nxmax-1,0:nymax-1,nuvzmax), vvh(0:nxmax-1,0:nymax-1,nuvzmax))
  allocate(pvh(0:nxmax-1,0:nymax-1,nuvzmax), wwh(0:nxmax-1,0:nymax-1,nwzmax))
  uuh=0.; vvh=0.; pvh=0.; wwh=0.
#ifdef FLEXREF_NESTS
  allocate(uuhn(0:nxmaxn-1,0:nymaxn-1,nuvzmax,maxnests), vvhn(0:nxmaxn-1,0:nymaxn-1,nuvzmax,maxnests))
  allocate(pvhn(0:nxmaxn-1,0:nymaxn-1,nuvzmax,maxnests), wwhn(0:nxmaxn-1,0:nymaxn-1,nwzmax,maxnests))
  uuhn=0.; vvhn=0.; pvhn=0.; wwhn=0.; gnxn=0; gnyn=0
#endif
  xglobal=.false.; nglobal=.false.; sglobal=.false.
  switchnorthg=999999.; switchsouthg=999999.
  readclouds=.false.; sumclouds=.false.; numbnests=0
  lsprec=0.; convprec=0.; ncalls=1
  gnx=0; gny=0; gnz=0

  open(uin, file=trim(fscen), access='stream', form='unformatted', status='old')
  do
    read(uin, iostat=ios) name, dtype, cnt
    if (ios .ne. 0) exit
    if (trim(name) .eq. 'END') exit
    n = int(cnt)
    if (dtype .eq. 1) then
      if (allocated(ibuf)) deallocate(ibuf)
      allocate(ibuf(n)); read(uin) ibuf
    else
      if (allocated(dbuf)) deallocate(dbuf)
      allocate(dbuf(n)); read(uin) dbuf
    end if
    select case (trim(name))
    case ('grid')      ! nx ny nz (= nuvz = nwz)
      gnx=ibuf(1); gny=ibuf(2); gnz=ibuf(3)
      if (gnx.gt.nxmax .or. gny.gt.nymax .or. gnz.gt.nzmax) stop 'grid too large'
      nx=gnx; ny=gny; nz=gnz; nuvz=gnz; nwz=gnz; nxfield=gnx
      nxmin1=nx-1; nymin1=ny-1
    case ('geom')      ! dx dy xlon0 ylat0
      dx=dbuf(1); dy=dbuf(2); xlon0=dbuf(3); ylat0=dbuf(4)
      dxconst=180./(dx*r_earth*pi)      ! as gridcheck_ecmwf.f90:311-312
      dyconst=180./(dy*r_earth*pi)
    case ('globalflags') ! xglobal nglobal sglobal
      xglobal=(ibuf(1).ne.0); nglobal=(ibuf(2).ne.0); sglobal=(ibuf(3).ne.0)
    case ('ncalls');  ncalls=ibuf(1)
    case ('akz');     akz(1:n)=dbuf(1:n)
    case ('bkz');     bkz(1:n)=dbuf(1:n)
    case ('aknew');   aknew(1:n)=dbuf(1:n)
    case ('bknew');   bknew(1:n)=dbuf(1:n)
    case ('ps');      call get2(ps(:,:,1,1))
    case ('tt2');     call get2(tt2(:,:,1,1))
    case ('td2');     call get2(td2(:,:,1,1))
    case ('tth');     call get3(tth(:,:,:,1), nuvzmax)
    case ('qvh');     call get3(qvh(:,:,:,1), nuvzmax)
    case ('uuh');     call get3(uuh, nuvzmax)
    case ('vvh');     call get3(vvh, nuvzmax)
    case ('pvh');     call get3(pvh, nuvzmax)
    case ('wwh');     call get3(wwh, nwzmax)
#ifdef FLEXREF_NESTS
    ! one nest (verttransform_nests.f90); geometry as gridcheck_nests.f90:359-372
    case ('nestgrid')
      gnxn=ibuf(1); gnyn=ibuf(2)
      if (gnxn.gt.nxmaxn .or. gnyn.gt.nymaxn) stop 'nest too large'
      numbnests=1; nxn(1)=gnxn; nyn(1)=gnyn
      call com_mod_allocate_nests
      uun=0.; vvn=0.; wwn=0.; ttn=0.; qvn=0.; pvn=0.; rhon=0.; drhodzn=0.; tthn=0.; qvhn=0.
      psn=0.; tt2n=0.; td2n=0.; lsprecn=0.; convprecn=0.
    case ('nestgeom')
      dxn(1)=dbuf(1); dyn(1)=dbuf(2); xlon0n(1)=dbuf(3); ylat0n(1)=dbuf(4)
      xresoln(1)=dx/dxn(1); yresoln(1)=dy/dyn(1)
    case ('psn');     call getn2(psn(:,:,1,1,1))
    case ('tt2n');    call getn2(tt2n(:,:,1,1,1))
    case ('td2n');    call getn2(td2n(:,:,1,1,1))
    case ('tthn');    call getn3(tthn(:,:,:,1,1), nuvzmax)
    case ('qvhn');    call getn3(qvhn(:,:,:,1,1), nuvzmax)
    case ('uuhn');    call getn3(uuhn(:,:,:,1), nuvzmax)
    case ('vvhn');    call getn3(vvhn(:,:,:,1), nuvzmax)
    case ('pvhn');    call getn3(pvhn(:,:,:,1), nuvzmax)
    case ('wwhn');    call getn3(wwhn(:,:,:,1), nwzmax)
#endif
    case default
      write(*,*) 'ref_vt_driver: unknown record ', trim(name)
      stop 1
    end select
  end do
  close(uin)

  ! Polar stereographic maps as the reference sets them up
  ! (call sequence of /root/reference/src/gridcheck_ecmwf.f90:341-366).
  if (sglobal) then
    sizesouth=6.*(switchsouth+90.)/dy
    call stlmbr(southpolemap,-90.,0.)
    call stcm2p(southpolemap,0.,0.,switchsouth,0.,sizesouth,sizesouth,switchsouth,180.)
    switchsouthg=(switchsouth-ylat0)/dy
  end if
  if (nglobal) then
    sizenorth=6.*(90.-switchnorth)/dy
    call stlmbr(northpolemap,90.,0.)
    call stcm2p(northpolemap,0.,0.,switchnorth,0.,sizenorth,sizenorth,switchnorth,180.)
    switchnorthg=(switchnorth-ylat0)/dy
  end if

  call system_clock(c0, crate)
  if (use_gpu .eq. 1) then
    ! defaults of the run switches flexgpu_init reads (no particle step is taken here)
    ldirect=1; lsynctime=900; method=1; mintime=1; ctl=0.2; ifine=4; turbswitch=.true.; cblflag=0
    mdomainfill=0; lsettling=.false.; nspec=1; DRYDEP=.false.; nageclass=1; lage(1)=999999999
    numpoint=1; allocate(xmass(1,maxspec), npart(1)); xmass=1.; npart(1)=1
    ipout=0; iflux=0; linit_cond=0; call com_mod_allocate_part(1)
    hmix(:,:,1,1)=500.; ustar(:,:,1,1)=0.3; wstar(:,:,1,1)=1.; oli(:,:,1,1)=0.01; tropopause(:,:,1,1)=10000.
    call flexgpu_init(gerr, nmaxpart=1, defer_height=.true.)
    if (gerr .ne. 0) then
      call flexgpu_last_error(gmsg); write(*,*) 'flexgpu_init: ', trim(gmsg); stop 1
    end if
    do icall=1,ncalls
      call flexgpu_verttransform(1,uuh,vvh,wwh,pvh,gerr)
      if (gerr .ne. 0) then
        call flexgpu_last_error(gmsg); write(*,*) 'flexgpu_verttransform: ', trim(gmsg); stop 1
      end if
    end do
#ifdef FLEXREF_NESTS
    if (numbnests .ge. 1) then
      xln(1)=(xlon0n(1)-xlon0)/dx; xrn(1)=(xlon0n(1)+real(nxn(1)-1)*dxn(1)-xlon0)/dx      ! gridcheck_nests.f90:362-372
      yln(1)=(ylat0n(1)-ylat0)/dy; yrn(1)=(ylat0n(1)+real(nyn(1)-1)*dyn(1)-ylat0)/dy
      hmixn=500.; ustarn=0.3; wstarn=1.; olin=0.01; tropopausen=10000.
      call flexgpu_nests_init(gerr)
      if (gerr .eq. 0) call flexgpu_verttransform_nests(1,uuhn,vvhn,wwhn,pvhn,gerr)
      if (gerr .ne. 0) then
        call flexgpu_last_error(gmsg); write(*,*) 'flexgpu_verttransform_nests: ', trim(gmsg); stop 1
      end if
    end if
#endif
  else
    do icall=1,ncalls
      call verttransform_ecmwf(1,uuh,vvh,wwh,pvh)
    end do
  end if
  call system_clock(c1)
#ifdef FLEXREF_NESTS
  if (numbnests .ge. 1 .and. use_gpu .eq. 0) call verttransform_nests(1,uuhn,vvhn,wwhn,pvhn)
#endif

  open(uout, file=trim(fout), access='stream', form='unformatted', status='replace')
  allocate(tmp(nz))
  tmp(1:nz)=height(1:nz)
  call put_d('height', tmp, nz)
  call put_i('nmixz', (/nmixz/), 1)
  call put_d('timing', (/real(c1-c0,kind=8)/real(crate,kind=8)/real(ncalls,kind=8)/), 1)
  call put_d('polemaps', (/real(northpolemap,kind=8), real(southpolemap,kind=8), real(switchnorthg,kind=8), real(switchsouthg,kind=8)/), 20)
  call dump3('uu', uu(:,:,:,1))
  call dump3('vv', vv(:,:,:,1))
  call dump3('ww', ww(:,:,:,1))
  call dump3('tt', tt(:,:,:,1))
  call dump3('qv', qv(:,:,:,1))
  call dump3('pv', pv(:,:,:,1))
  call dump3('rho', rho(:,:,:,1))
  call dump3('drhodz', drhodz(:,:,:,1))
  if (nglobal .or. sglobal) then
    call dump3('uupol', uupol(:,:,:,1))
    call dump3('vvpol', vvpol(:,:,:,1))
  end if
#ifdef FLEXREF_NESTS
  if (numbnests .ge. 1) then
    call dumpn3('uun', uun(:,:,:,1,1))
    call dumpn3('vvn', vvn(:,:,:,1,1))
    call dumpn3('wwn', wwn(:,:,:,1,1))
    call dumpn3('ttn', ttn(:,:,:,1,1))
    call dumpn3('qvn', qvn(:,:,:,1,1))
    call dumpn3('pvn', pvn(:,:,:,1,1))
    call dumpn3('rhon', rhon(:,:,:,1,1))
    call dumpn3('drhodzn', drhodzn(:,:,:,1,1))
  end if
#endif
  write(uout) 'END             ', 1_4, 0_8
  close(uout)

contains
#ifdef FLEXREF_NESTS
  subroutine getn2(a)
    real, intent(inout) :: a(0:nxmaxn-1,0:nymaxn-1)
    integer :: ix, jy
    a=0.
    do jy=0,gnyn-1
      do ix=0,gnxn-1
        a(ix,jy)=dbuf(1+ix+gnxn*jy)
      end do
    end do
  end subroutine getn2
  subroutine getn3(a, nl)
    integer, intent(in) :: nl
    real, intent(inout) :: a(0:nxmaxn-1,0:nymaxn-1,nl)
    integer :: ix, jy, k
    a=0.
    do k=1,gnz
      do jy=0,gnyn-1
        do ix=0,gnxn-1
          a(ix,jy,k)=dbuf(1+ix+gnxn*(jy+gnyn*(k-1)))
        end do
      end do
    end do
  end subroutine getn3
  subroutine dumpn3(nm, a)
    character(len=*), intent(in) :: nm
    real, intent(in) :: a(0:nxmaxn-1,0:nymaxn-1,nzmax)
    real(kind=8), allocatable :: t(:)
    integer :: ix, jy, k
    allocate(t(gnxn*gnyn*gnz))
    do k=1,gnz
      do jy=0,gnyn-1
        do ix=0,gnxn-1
          t(1+ix+gnxn*(jy+gnyn*(k-1)))=a(ix,jy,k)
        end do
      end do
    end do
    call put_d(nm, t, gnxn*gnyn*gnz)
    deallocate(t)
  end subroutine dumpn3
#endif
  subroutine get2(a)
    real, intent(inout) :: a(0:nxmax-1,0:nymax-1)
    integer :: ix, jy
    a=0.
    do jy=0,gny-1
      do ix=0,gnx-1
        a(ix,jy)=dbuf(1+ix+gnx*jy)
      end do
    end do
  end subroutine get2
  subroutine get3(a, nl)
    integer, intent(in) :: nl
    real, intent(inout) :: a(0:nxmax-1,0:nymax-1,nl)
    integer :: ix, jy, k
    a=0.
    do k=1,gnz
      do jy=0,gny-1
        do ix=0,gnx-1
          a(ix,jy,k)=dbuf(1+ix+gnx*(jy+gny*(k-1)))
        end do
      end do
    end do
  end subroutine get3
  subroutine dump3(nm, a)
    character(len=*), intent(in) :: nm
    real, intent(in) :: a(0:nxmax-1,0:nymax-1,nzmax)
    real(kind=8), allocatable :: t(:)
    integer :: ix, jy, k
    allocate(t(gnx*gny*gnz))
    do k=1,gnz
      do jy=0,gny-1
        do ix=0,gnx-1
          t(1+ix+gnx*(jy+gny*(k-1)))=a(ix,jy,k)
        end do
      end do
    end do
    call put_d(nm, t, gnx*gny*gnz)
    deallocate(t)
  end subroutine dump3
end program vtref
