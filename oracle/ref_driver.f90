! TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
!
! ref_driver: drives the *unmodified* reference FLEXPART hot path
! (initialize.f90 / advance.f90 and their helpers, compiled where they lie
! under /root/reference/src by oracle/build_ref.sh) from a scenario file, in
! exactly the order of the reference's serial particle loop
! (timemanager.f90:531-712), and dumps the particle SoA after every
! synchronisation step.  This file is our own code: it contains no reference
! source, only calls into it and assignments to its module variables.
!
! Usage:  flexref_rK scenario.bin out.bin [timing|gpu|gpu32]
!   gpu: the same host arrays (com_mod) are advanced by the MI355X engine through the
!        ISO_C_BINDING shim flexpart_amd/fortran/flexgpu_mod.f90 instead of the Fortran loop --
!        the drop-in integration test (needs a GPU; run on the GPU box).
!
! Scenario file = sequence of records {name*16, dtype i4 (1=i32, 2=f64),
! count i8, payload}, terminated by name 'END'.  All reals travel as f64 and
! are converted to this build's default real kind (4 or 8) on assignment.

module drv_io
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
end module drv_io

program flexref
  use par_mod
  use com_mod
  use point_mod
  use interpol_mod
  use hanna_mod
  use cmapf_mod
  use random_mod
  use unc_mod
  use outg_mod
  use drv_io
  use flexgpu_mod
  implicit none

  type(fpx_step_stats) :: gstats
  integer :: gerr, use_gpu
  character(len=256) :: gmsg
  character(len=512) :: fscen, fout, arg3
  character(len=16) :: name
  integer(kind=4) :: dtype
  integer(kind=8) :: cnt
  integer, allocatable :: ibuf(:)
  real(kind=8), allocatable :: dbuf(:)
  integer :: ios, i, j, k, m, ks, n, idummy, istep, nsteps, itime0, itime
  integer :: npart_in, gnx, gny, gnz, nstop, timing, do_conc, do_concn, nage, kp
  integer :: ldeltat, loutnext_d, itage
  integer(kind=8) :: c0, c1, crate, nadv
  real :: prob(maxspec), decfact, xmassfract, weight
  real :: prob_rec(maxspec), grfraction(3), wetscav                    ! as timemanager.f90:105,109
  integer(selected_int_kind(16)), dimension(maxspec) :: idummy1, idummy2
  real(dep_prec) :: drydeposit(maxspec)      ! as timemanager.f90:104
  real :: sizenorth, sizesouth
  real(kind=8), allocatable :: tmp(:)
  logical :: have_pol, do_polar_setup
  real(kind=8) :: tsec

  call get_command_argument(1, fscen)
  call get_command_argument(2, fout)
  timing = 0
  use_gpu = 0
  if (command_argument_count() .ge. 3) then
    call get_command_argument(3, arg3)
    if (trim(arg3) .eq. 'timing') timing = 1
    if (trim(arg3) .eq. 'gpu') use_gpu = 1
    if (trim(arg3) .eq. 'gpu32') use_gpu = 2      ! the reference-typed f32 engine (compute_real_bytes = 4)
  end if

  ! Random number table exactly as the reference main program fills it
  ! (call sequence of /root/reference/src/FLEXPART.f90:47,56-59).
  idummy = -320
  do i=1,maxrand-1,2
    call gasdev1(idummy,rannumb(i),rannumb(i+1))
  end do
  call gasdev1(idummy,rannumb(maxrand),rannumb(maxrand-1))

  ! ---- defaults ----------------------------------------------------------
  ipout=0; iflux=0; ldirect=1; lsynctime=900; method=1; mintime=1; ctl=0.2; ifine=4
  fine=0.25; turbswitch=.true.; cblflag=0; mdomainfill=0; mquasilag=0
  lsettling=.false.; nspec=1; maxpointspec_act=1
  DRYDEP=.false.; WETDEP=.false.; DRYBKDEP=.false.; WETBKDEP=.false.
  idummy1(:)=0; idummy2(:)=0
  DRYDEPSPEC(:)=.false.; WETDEPSPEC(:)=.false.
  density(:)=0.; decay(:)=0.; dquer(:)=0.; vsetaver(:)=0.; cunningham(:)=1.
  numbnests=0; nageclass=1; lage(1)=999999999
  ioutputforeachrelease=0; nested_output=0; linit_cond=0; ind_samp=0
  xglobal=.false.; nglobal=.false.; sglobal=.false.
  switchnorthg=999999.; switchsouthg=999999.
  nsteps=1; itime0=0; nan_count=0; nan_count2=0
  memind(1)=1; memind(2)=2; memind(3)=3
  numpoint=1; have_pol=.false.; do_polar_setup=.false.; do_conc=0; do_concn=0; numreceptor=0
  numreceptor=0; loutnext_d=0
  gnx=0; gny=0; gnz=0; npart_in=0

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
    ! --- grid --------------------------------------------------------------
    case ('grid')      ! nx ny nz
      gnx=ibuf(1); gny=ibuf(2); gnz=ibuf(3)
      if (gnx.gt.nxmax .or. gny.gt.nymax .or. gnz.gt.nzmax) stop 'grid too large'
      nx=gnx; ny=gny; nz=gnz; nuvz=gnz; nwz=gnz; nxfield=gnx
      nxmin1=nx-1; nymin1=ny-1
    case ('geom')      ! dx dy xlon0 ylat0
      dx=dbuf(1); dy=dbuf(2); xlon0=dbuf(3); ylat0=dbuf(4)
      ! as /root/reference/src/gridcheck_ecmwf.f90:311-312
      dxconst=180./(dx*r_earth*pi)
      dyconst=180./(dy*r_earth*pi)
    case ('globalflags') ! xglobal nglobal sglobal
      xglobal=(ibuf(1).ne.0); nglobal=(ibuf(2).ne.0); sglobal=(ibuf(3).ne.0)
      do_polar_setup = nglobal .or. sglobal
    case ('height');   height(1:n)=dbuf(1:n)
    case ('nmixz');    nmixz=ibuf(1)
    case ('memtime');  memtime(1)=ibuf(1); memtime(2)=ibuf(2)
      lwindinterv=abs(memtime(2)-memtime(1))
    case ('memind');   memind(1)=ibuf(1); memind(2)=ibuf(2)
    ! --- run switches --------------------------------------------------------
    case ('ldirect');  ldirect=ibuf(1)
    case ('lsynctime'); lsynctime=ibuf(1)
    case ('method');   method=ibuf(1)
    case ('mintime');  mintime=ibuf(1)
    case ('ctl');      ctl=dbuf(1)
    case ('ifine');    ifine=ibuf(1); fine=1./real(ifine)
    case ('turbswitch'); turbswitch=(ibuf(1).ne.0)
    case ('cblflag');  cblflag=ibuf(1)
    case ('mdomainfill'); mdomainfill=ibuf(1)
    case ('lsettling'); lsettling=(ibuf(1).ne.0)
    case ('nspec');    nspec=ibuf(1)
    case ('drydep');   DRYDEP=(ibuf(1).ne.0)
    case ('drydepspec'); do i=1,n; DRYDEPSPEC(i)=(ibuf(i).ne.0); end do
    case ('density');  density(1:n)=dbuf(1:n)
    case ('dquer');    dquer(1:n)=dbuf(1:n)
    case ('vsetaver'); vsetaver(1:n)=dbuf(1:n)
    case ('cunningham'); cunningham(1:n)=dbuf(1:n)
    case ('decay');    decay(1:n)=dbuf(1:n)
    case ('turbpar')   ! d_trop d_strat turbmesoscale
      d_trop=dbuf(1); d_strat=dbuf(2); turbmesoscale=dbuf(3)
    case ('lage')
      if (n.gt.maxageclass) stop 'more age classes than this build holds (par_mod maxageclass)'
      nageclass=n; lage(1:n)=ibuf(1:n)
    case ('nsteps');   nsteps=ibuf(1)
    case ('mquasilag'); mquasilag=ibuf(1)
    case ('numpoint');  numpoint=ibuf(1)          ! before 'npart': xmass(numpoint,maxspec), npart(numpoint)
    case ('nclassunc')                            ! a compile-time size of the reference (par_mod): must match this build
      if (ibuf(1).gt.nclassunc) stop 'nclassunc of the scenario exceeds this build'
    ! --- output grid (readoutgrid.f90 / outgrid_init.f90 state) ---------------------------
    case ('outgrid')   ! numxgrid numygrid numzgrid
      numxgrid=ibuf(1); numygrid=ibuf(2); numzgrid=ibuf(3); do_conc=1
    case ('outgeom')   ! dxout dyout outlon0 outlat0
      dxout=dbuf(1); dyout=dbuf(2); outlon0=dbuf(3); outlat0=dbuf(4)
    case ('outheight')
      allocate(outheight(n), outheighthalf(n)); outheight(1:n)=dbuf(1:n)
    ! --- nested output grid (readoutgrid_nest.f90 state) and receptor points (readreceptors.f90) ---
    case ('outgridn')  ! numxgridn numygridn
      numxgridn=ibuf(1); numygridn=ibuf(2); do_concn=1
    case ('outgeomn')  ! dxoutn dyoutn outlon0n outlat0n
      dxoutn=dbuf(1); dyoutn=dbuf(2); outlon0n=dbuf(3); outlat0n=dbuf(4)
    case ('receptors') ! xreceptor(1:m), yreceptor(1:m), receptorarea(1:m) in grid coordinates / m2
      numreceptor=int(n/3)
      xreceptor(1:numreceptor)=dbuf(1:numreceptor)
      yreceptor(1:numreceptor)=dbuf(numreceptor+1:2*numreceptor)
      receptorarea(1:numreceptor)=dbuf(2*numreceptor+1:3*numreceptor)
    case ('concflags') ! ind_samp ioutputforeachrelease
      ind_samp=ibuf(1); ioutputforeachrelease=ibuf(2)
    case ('outtimes')  ! loutnext loutstep
      loutnext_d=ibuf(1); loutstep=ibuf(2)
    ! --- wet deposition (species parameters of readspecies.f90, precipitation/cloud fields) ---
    case ('wetdep');     WETDEP=(ibuf(1).ne.0)
    case ('wetdepspec'); do i=1,n; WETDEPSPEC(i)=(ibuf(i).ne.0); end do
    case ('weta_gas');   weta_gas(1:n)=dbuf(1:n)
    case ('wetb_gas');   wetb_gas(1:n)=dbuf(1:n)
    case ('crain_aero'); crain_aero(1:n)=dbuf(1:n)
    case ('csnow_aero'); csnow_aero(1:n)=dbuf(1:n)
    case ('ccn_aero');   ccn_aero(1:n)=dbuf(1:n)
    case ('in_aero');    in_aero(1:n)=dbuf(1:n)
    case ('henry');      henry(1:n)=dbuf(1:n)
    case ('lsprec');     call fill2(lsprec, dbuf)
    case ('convprec');   call fill2(convprec, dbuf)
    case ('tcc');        call fill2(tcc, dbuf)
    case ('clouds')      ! compact (nx,ny,nz,2) int
      do m=1,2
        do k=1,gnz
          do j=0,gny-1
            do i=0,gnx-1
              clouds(i,j,k,m)=int(ibuf(1+i+gnx*(j+gny*((k-1)+gnz*(m-1)))),1)
            end do
          end do
        end do
      end do
    case ('cloudsh')     ! compact (nx,ny,2) int
      do m=1,2
        do j=0,gny-1
          do i=0,gnx-1
            cloudsh(i,j,m)=ibuf(1+i+gnx*(j+gny*(m-1)))
          end do
        end do
      end do
    case ('itime0');   itime0=ibuf(1)
    ! --- 3-D fields: compact (nx,ny,nz,2), x fastest -------------------------
    case ('uu');     call fill3(uu, dbuf)
    case ('vv');     call fill3(vv, dbuf)
    case ('ww');     call fill3(ww, dbuf)
    case ('rho');    call fill3(rho, dbuf)
    case ('drhodz'); call fill3(drhodz, dbuf)
    case ('tt');     call fill3(tt, dbuf)
    case ('uupol');  call fill3(uupol, dbuf); have_pol=.true.
    case ('vvpol');  call fill3(vvpol, dbuf); have_pol=.true.
#ifdef FLEXREF_NESTS
    ! --- one nested grid (needs a par_mod with maxnests >= 1: the *n build variants) ----------
    case ('nest')      ! nxn nyn
      numbnests=1; nxn(1)=ibuf(1); nyn(1)=ibuf(2)
      if (nxn(1).gt.nxmaxn .or. nyn(1).gt.nymaxn) stop 'nest too large'
      call com_mod_allocate_nests
      readclouds_nest=.false.; sumclouds_nest=.false.
    case ('nestgeom')  ! dxn dyn xlon0n ylat0n ; derived geometry as gridcheck_nests.f90:362-378
      dxn(1)=dbuf(1); dyn(1)=dbuf(2); xlon0n(1)=dbuf(3); ylat0n(1)=dbuf(4)
      xresoln(0)=1.; yresoln(0)=1.
      xresoln(1)=dx/dxn(1)
      yresoln(1)=dy/dyn(1)
      xln(1)=(xlon0n(1)-xlon0)/dx
      xrn(1)=(xlon0n(1)+real(nxn(1)-1)*dxn(1)-xlon0)/dx
      yln(1)=(ylat0n(1)-ylat0)/dy
      yrn(1)=(ylat0n(1)+real(nyn(1)-1)*dyn(1)-ylat0)/dy
    case ('uun');     call fill3n(uun, dbuf)
    case ('vvn');     call fill3n(vvn, dbuf)
    case ('wwn');     call fill3n(wwn, dbuf)
    case ('rhon');    call fill3n(rhon, dbuf)
    case ('drhodzn'); call fill3n(drhodzn, dbuf)
    case ('hmixn');   call fill2n(hmixn, dbuf)
    case ('ustarn');  call fill2n(ustarn, dbuf)
    case ('wstarn');  call fill2n(wstarn, dbuf)
    case ('olin');    call fill2n(olin, dbuf)
    case ('tropopausen'); call fill2n(tropopausen, dbuf)
    ! precipitation / cloud / temperature fields of the nest (get_wetscav.f90:126-128,150-151,197-199)
    case ('lsprecn');   call fill2n(lsprecn, dbuf)
    case ('convprecn'); call fill2n(convprecn, dbuf)
    case ('tccn');      call fill2n(tccn, dbuf)
    case ('ttn');       call fill3n(ttn, dbuf)
    case ('cloudsn')    ! compact (nxn,nyn,nz,2) int
      do m=1,2
        do k=1,gnz
          do j=0,nyn(1)-1
            do i=0,nxn(1)-1
              cloudsn(i,j,k,m,1)=int(ibuf(1+i+nxn(1)*(j+nyn(1)*((k-1)+gnz*(m-1)))),1)
            end do
          end do
        end do
      end do
    case ('cloudshn')   ! compact (nxn,nyn,2) int
      do m=1,2
        do j=0,nyn(1)-1
          do i=0,nxn(1)-1
            cloudshn(i,j,m,1)=ibuf(1+i+nxn(1)*(j+nyn(1)*(m-1)))
          end do
        end do
      end do
    case ('vdepn')    ! compact (nxn,nyn,nspec,2)
      do m=1,2
        do ks=1,nspec
          do j=0,nyn(1)-1
            do i=0,nxn(1)-1
              vdepn(i,j,ks,m,1)=dbuf(1+i+nxn(1)*(j+nyn(1)*((ks-1)+nspec*(m-1))))
            end do
          end do
        end do
      end do
#endif
    ! --- 2-D fields: compact (nx,ny,2) -----------------------------------------
    case ('hmix');   call fill2(hmix, dbuf)
    case ('ustar');  call fill2(ustar, dbuf)
    case ('wstar');  call fill2(wstar, dbuf)
    case ('oli');    call fill2(oli, dbuf)
    case ('tropopause'); call fill2(tropopause, dbuf)
    case ('vdep')    ! compact (nx,ny,nspec,2)
      do m=1,2
        do ks=1,nspec
          do j=0,gny-1
            do i=0,gnx-1
              vdep(i,j,ks,m)=dbuf(1+i+gnx*(j+gny*((ks-1)+nspec*(m-1))))
            end do
          end do
        end do
      end do
    ! --- particles -------------------------------------------------------------
    case ('npart')
      npart_in=ibuf(1); numpart=npart_in
      call com_mod_allocate_part(npart_in)
      allocate(xmass(numpoint,maxspec), npart(numpoint))
      xmass(:,:)=1.; npart(:)=npart_in
      itra1(:)=-999999999; npoint(:)=1; nclass(:)=1; idt(:)=0; itramem(:)=0
      itrasplit(:)=999999999; xmass1(:,:)=0.
      uap(:)=0.; ucp(:)=0.; uzp(:)=0.; us(:)=0.; vs(:)=0.; ws(:)=0.; cbt(:)=1
    ! --- backward runs with receptor scavenging (readcommand.f90:320-340, readreleases.f90:508-517) ------
    case ('drybkdep')
      DRYBKDEP=(ibuf(1).ne.0)
      if (DRYBKDEP .and. .not. allocated(xscav_frac1)) then
        allocate(xscav_frac1(npart_in,maxspec)); xscav_frac1(:,:)=-1.      ! releaseparticles.f90:167-171
      end if
    case ('wetbkdep')
      WETBKDEP=(ibuf(1).ne.0)
      if (WETBKDEP .and. .not. allocated(xscav_frac1)) then
        allocate(xscav_frac1(npart_in,maxspec)); xscav_frac1(:,:)=-1.
      end if
    case ('zpoint1');  if (.not. allocated(zpoint1)) allocate(zpoint1(numpoint)); zpoint1(1:n)=dbuf(1:n)
    case ('zpoint2');  if (.not. allocated(zpoint2)) allocate(zpoint2(numpoint)); zpoint2(1:n)=dbuf(1:n)
    case ('xtra1');   xtra1(1:n)=dbuf(1:n)
    case ('ytra1');   ytra1(1:n)=dbuf(1:n)
    case ('ztra1');   ztra1(1:n)=dbuf(1:n)
    case ('itra1');   itra1(1:n)=ibuf(1:n)
    case ('itramem'); itramem(1:n)=ibuf(1:n)
    case ('npoint');  npoint(1:n)=ibuf(1:n)
    case ('nclass');  nclass(1:n)=ibuf(1:n)
    case ('idt');     idt(1:n)=ibuf(1:n)
    case ('uap');     uap(1:n)=dbuf(1:n)
    case ('ucp');     ucp(1:n)=dbuf(1:n)
    case ('uzp');     uzp(1:n)=dbuf(1:n)
    case ('us');      us(1:n)=dbuf(1:n)
    case ('vs');      vs(1:n)=dbuf(1:n)
    case ('ws');      ws(1:n)=dbuf(1:n)
    case ('cbt');     do i=1,n; cbt(i)=int(ibuf(i),2); end do
    case ('xmass1')   ! (npart, nspec) species-major
      do ks=1,nspec
        xmass1(1:npart_in,ks)=dbuf(1+(ks-1)*npart_in:ks*npart_in)
      end do
    case ('xmass')    ! release masses xmass(numpoint,nspec), species-major like the Fortran array
      do ks=1,nspec
        xmass(1:numpoint,ks)=dbuf(1+(ks-1)*numpoint:ks*numpoint)
      end do
    case ('npart_rel') ! npart(numpoint)
      npart(1:numpoint)=ibuf(1:numpoint)
    case default
      write(*,*) 'ref_driver: unknown record ', trim(name)
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

  if (do_conc .eq. 1) then
    xoutshift=xlon0-outlon0          ! readoutgrid.f90:199-200
    youtshift=ylat0-outlat0
    maxpointspec_act=1
    if (ioutputforeachrelease.eq.1) maxpointspec_act=numpoint
    ! allocation as outgrid_init.f90:192-200
    allocate(gridunc(0:numxgrid-1,0:numygrid-1,numzgrid,maxspec,maxpointspec_act,nclassunc,maxageclass))
    allocate(drygridunc(0:numxgrid-1,0:numygrid-1,maxspec,maxpointspec_act,nclassunc,maxageclass))
    allocate(wetgridunc(0:numxgrid-1,0:numygrid-1,maxspec,maxpointspec_act,nclassunc,maxageclass))
    gridunc=0.; drygridunc=0.; wetgridunc=0.
    creceptor=0.
    if (do_concn .eq. 1) then
      nested_output=1
      xoutshiftn=xlon0-outlon0n      ! readoutgrid_nest.f90
      youtshiftn=ylat0-outlat0n
      allocate(griduncn(0:numxgridn-1,0:numygridn-1,numzgrid,maxspec,maxpointspec_act,nclassunc,maxageclass))
      allocate(drygriduncn(0:numxgridn-1,0:numygridn-1,maxspec,maxpointspec_act,nclassunc,maxageclass))
      allocate(wetgriduncn(0:numxgridn-1,0:numygridn-1,maxspec,maxpointspec_act,nclassunc,maxageclass))
      griduncn=0.; drygriduncn=0.; wetgriduncn=0.
    end if
  end if

  open(uout, file=trim(fout), access='stream', form='unformatted', status='replace')
  allocate(tmp(max(npart_in,maxrand)))

  if (timing .eq. 0) then
    tmp(1:maxrand)=rannumb(1:maxrand)
    call put_d('rannumb', tmp, maxrand)
    tmp(1:9)=northpolemap(1:9); call put_d('northpolemap', tmp, 9)
    tmp(1:9)=southpolemap(1:9); call put_d('southpolemap', tmp, 9)
    tmp(1)=switchnorthg; tmp(2)=switchsouthg; tmp(3)=dxconst; tmp(4)=dyconst
    call put_d('derived', tmp, 4)
  end if

  if (use_gpu .ge. 1) then
    ! ---- drop-in: the engine replaces the particle loop -------------------------------
    if (use_gpu .eq. 2) then
      call flexgpu_init(gerr, nmaxpart=numpart, compute_real_bytes=4)
    else
      call flexgpu_init(gerr, nmaxpart=numpart)
    end if
    if (gerr .ne. 0) call gpu_fail('flexgpu_init')
    call flexgpu_use_table_rng(gerr)
    if (gerr .ne. 0) call gpu_fail('flexgpu_use_table_rng')
    call flexgpu_upload_fields(memind(1), gerr)
    if (gerr .ne. 0) call gpu_fail('flexgpu_upload_fields 1')
    call flexgpu_upload_fields(memind(2), gerr)
    if (gerr .ne. 0) call gpu_fail('flexgpu_upload_fields 2')
#ifdef FLEXREF_NESTS
    if (numbnests .gt. 0) then
      call flexgpu_upload_nests(gerr)
      if (gerr .ne. 0) call gpu_fail('flexgpu_upload_nests')
    end if
#endif
    call flexgpu_set_windtime(gerr)
    if (gerr .ne. 0) call gpu_fail('flexgpu_set_windtime')
    call flexgpu_upload_particles(1, numpart, gerr)
    if (gerr .ne. 0) call gpu_fail('flexgpu_upload_particles')
    if (do_conc .eq. 1) then
      call flexgpu_outgrid_init(loutnext_d, gerr)
      if (gerr .ne. 0) call gpu_fail('flexgpu_outgrid_init')
      if (WETDEP) then
        call flexgpu_wet_init(gerr)
        if (gerr .ne. 0) call gpu_fail('flexgpu_wet_init')
        call flexgpu_upload_wet_fields(memind(1), gerr)
        if (gerr .ne. 0) call gpu_fail('flexgpu_upload_wet_fields 1')
        call flexgpu_upload_wet_fields(memind(2), gerr)
        if (gerr .ne. 0) call gpu_fail('flexgpu_upload_wet_fields 2')
#ifdef FLEXREF_NESTS
        if (numbnests .gt. 0) then
          call flexgpu_upload_wet_nest_fields(memind(1), gerr)
          if (gerr .ne. 0) call gpu_fail('flexgpu_upload_wet_nest_fields 1')
          call flexgpu_upload_wet_nest_fields(memind(2), gerr)
          if (gerr .ne. 0) call gpu_fail('flexgpu_upload_wet_nest_fields 2')
        end if
#endif
      end if
    end if
    nadv=0
    call system_clock(c0, crate)
    do istep=0,nsteps-1
      itime=itime0+istep*lsynctime
      if (WETDEP .and. (do_conc .eq. 1) .and. itime .ne. 0 .and. numpart .gt. 0) then   ! timemanager.f90:164-169
        call flexgpu_wetdepo(itime, lsynctime, loutnext_d, gerr)
        if (gerr .ne. 0) call gpu_fail('flexgpu_wetdepo')
      end if
      call flexgpu_step(itime, gstats, gerr)
      if (gerr .ne. 0) call gpu_fail('flexgpu_step')
      if (do_conc .eq. 1) then
        call flexgpu_conccalc(itime+lsynctime, 1.0, gerr)
        if (gerr .ne. 0) call gpu_fail('flexgpu_conccalc')
      end if
      nadv=nadv+gstats%n_due
      call flexgpu_download_particles(1, numpart, gerr)
      if (gerr .ne. 0) call gpu_fail('flexgpu_download_particles')
      call dump_state()
    end do
    call system_clock(c1)
    if (do_conc .eq. 1) then
      call flexgpu_get_grids(0, gerr)        ! into the host's own gridunc, drygridunc, wetgridunc (+ nested, creceptor)
      if (gerr .ne. 0) call gpu_fail('flexgpu_get_grids')
      call dump_grids()
    end if
    call flexgpu_finalize()
    goto 900
  end if

  ! ---- the particle loop, in the order of timemanager.f90:531-712 ---------
  nadv=0
  call system_clock(c0, crate)
  do istep=0,nsteps-1
    itime=itime0+istep*lsynctime
    ! wet deposition first, as timemanager.f90:164-169
    if (WETDEP .and. itime .ne. 0 .and. numpart .gt. 0) call wetdepo(itime,lsynctime,loutnext_d)
    if (itime.lt.loutnext_d) then       ! timemanager.f90:513-517
      ldeltat=itime-(loutnext_d-loutstep)
    else
      ldeltat=itime-loutnext_d
    endif
    do j=1,numpart
      if (itra1(j).eq.itime) then
        if (ioutputforeachrelease.eq.1) then      ! timemanager.f90:538-542
          kp=npoint(j)
        else
          kp=1
        endif
        itage=abs(itra1(j)-itramem(j))
        do nage=1,nageclass
          if (itage.lt.lage(nage)) exit
        end do
        if ((itramem(j).eq.itime).or.(itime.eq.0)) &
             call initialize(itime,idt(j),uap(j),ucp(j),uzp(j), &
             us(j),vs(j),ws(j),xtra1(j),ytra1(j),ztra1(j),cbt(j))
        ! RECEPTOR: dry/wet depovel -- the statements of timemanager.f90:571-598, calling the reference's own
        ! get_vdep_prob and get_wetscav
        if  (DRYBKDEP) then
          do ks=1,nspec
            if  ((xscav_frac1(j,ks).lt.0)) then
              call get_vdep_prob(itime,xtra1(j),ytra1(j),ztra1(j),prob_rec)
              if (DRYDEPSPEC(ks)) then
                xscav_frac1(j,ks)=prob_rec(ks)
              else
                xmass1(j,ks)=0.
                xscav_frac1(j,ks)=0.
              endif
            endif
          enddo
        endif
        if (WETBKDEP) then
          do ks=1,nspec
            if  ((xscav_frac1(j,ks).lt.0)) then
              call get_wetscav(itime,lsynctime,loutnext_d,j,ks,grfraction,idummy1,idummy2,wetscav)
              if (wetscav.gt.0) then
                xscav_frac1(j,ks)=wetscav* &
                     (zpoint2(npoint(j))-zpoint1(npoint(j)))*grfraction(1)
              else
                xmass1(j,ks)=0.
                xscav_frac1(j,ks)=0.
              endif
            endif
          enddo
        endif
        call advance(itime,npoint(j),idt(j),uap(j),ucp(j),uzp(j), &
             us(j),vs(j),ws(j),nstop,xtra1(j),ytra1(j),ztra1(j),prob, &
             cbt(j))
        nadv=nadv+1
        ! epilogue: our restatement of timemanager.f90:630-708 (mass update,
        ! minmass and age termination); the deposition-grid kernels are driven
        ! separately.
        if (nstop.gt.1) then
          itra1(j)=-999999999
        else
          itra1(j)=itime+lsynctime
          xmassfract=0.
          do ks=1,nspec
            if (decay(ks).gt.0.) then
              decfact=exp(-real(abs(lsynctime))*decay(ks))
            else
              decfact=1.
            endif
            if (DRYDEPSPEC(ks)) then
              drydeposit(ks)=xmass1(j,ks)*prob(ks)*decfact
              xmass1(j,ks)=xmass1(j,ks)*(1.-prob(ks))*decfact
              if (decay(ks).gt.0.) then
                drydeposit(ks)=drydeposit(ks)*exp(real(abs(ldeltat))*decay(ks))
              endif
            else
              xmass1(j,ks)=xmass1(j,ks)*decfact
            endif
            if (mdomainfill.eq.0.and.mquasilag.eq.0) then
              if (xmass(npoint(j),ks).gt.0.) &
                   xmassfract=max(xmassfract,real(npart(npoint(j)))* &
                   xmass1(j,ks)/xmass(npoint(j),ks))
            else
              xmassfract=1.0
            end if
          end do
          if (xmassfract.lt.minmass) itra1(j)=-999999999
          if (DRYDEP.and.(ldirect.eq.1).and.(do_conc.eq.1)) &
               call drydepokernel(nclass(j),drydeposit,real(xtra1(j)),real(ytra1(j)),nage,kp)
          if (DRYDEP.and.(ldirect.eq.1).and.(do_concn.eq.1)) &      ! timemanager.f90:694-696
               call drydepokernel_nest(nclass(j),drydeposit,real(xtra1(j)),real(ytra1(j)),nage,kp)
          if (abs(itra1(j)-itramem(j)).ge.lage(nageclass)) itra1(j)=-999999999
        endif
      endif
    end do
    ! sample the particles at their new positions (conccalc.f90; the time manager does this at
    ! the top of the next iteration, timemanager.f90:350-365)
    if (do_conc .eq. 1) call conccalc(itime+lsynctime, 1.0)
    if (timing .eq. 0) call dump_state()
  end do
  call system_clock(c1)
900 continue
  tsec = real(c1-c0,8)/real(crate,8)

  if (do_conc .eq. 1 .and. use_gpu .eq. 0) call dump_grids()
  tmp(1)=tsec; tmp(2)=real(nadv,8); tmp(3)=real(nan_count,8); tmp(4)=real(nan_count2,8)
  call put_d('timing', tmp, 4)
  name='END'
  write(uout) name, 1_4, 0_8
  close(uout)
  write(*,'(a,f10.4,a,i12,a,es12.4)') 'ref_driver: particle loop ', tsec, ' s, ', nadv, &
       ' advance calls, particle-steps/s = ', real(nadv,8)/max(tsec,1d-9)

contains

  subroutine gpu_fail(where)
    character(len=*), intent(in) :: where
    call flexgpu_last_error(gmsg)
    write(*,*) 'ref_driver: ', where, ' failed: ', gerr, ' ', trim(gmsg)
    stop 2
  end subroutine gpu_fail

  subroutine fill3(f, b)
    real, intent(inout) :: f(0:nxmax-1,0:nymax-1,nzmax,numwfmem)
    real(kind=8), intent(in) :: b(*)
    integer :: ii,jj,kk,mm
    do mm=1,2
      do kk=1,gnz
        do jj=0,gny-1
          do ii=0,gnx-1
            f(ii,jj,kk,mm)=b(1+ii+gnx*(jj+gny*((kk-1)+gnz*(mm-1))))
          end do
        end do
      end do
    end do
  end subroutine fill3

#ifdef FLEXREF_NESTS
  subroutine fill3n(f, b)
    real, intent(inout) :: f(0:nxmaxn-1,0:nymaxn-1,nzmax,numwfmem,*)
    real(kind=8), intent(in) :: b(*)
    integer :: ii,jj,kk,mm
    do mm=1,2
      do kk=1,gnz
        do jj=0,nyn(1)-1
          do ii=0,nxn(1)-1
            f(ii,jj,kk,mm,1)=b(1+ii+nxn(1)*(jj+nyn(1)*((kk-1)+gnz*(mm-1))))
          end do
        end do
      end do
    end do
  end subroutine fill3n

  subroutine fill2n(f, b)
    real, intent(inout) :: f(0:nxmaxn-1,0:nymaxn-1,1,numwfmem,*)
    real(kind=8), intent(in) :: b(*)
    integer :: ii,jj,mm
    do mm=1,2
      do jj=0,nyn(1)-1
        do ii=0,nxn(1)-1
          f(ii,jj,1,mm,1)=b(1+ii+nxn(1)*(jj+nyn(1)*(mm-1)))
        end do
      end do
    end do
  end subroutine fill2n
#endif

  subroutine fill2(f, b)
    real, intent(inout) :: f(0:nxmax-1,0:nymax-1,1,numwfmem)
    real(kind=8), intent(in) :: b(*)
    integer :: ii,jj,mm
    do mm=1,2
      do jj=0,gny-1
        do ii=0,gnx-1
          f(ii,jj,1,mm)=b(1+ii+gnx*(jj+gny*(mm-1)))
        end do
      end do
    end do
  end subroutine fill2

  subroutine dump_grids()
    real(kind=8), allocatable :: g(:)
    integer :: ng
    ! the last dimension is allocated maxageclass (outgrid_init.f90:192): the nageclass planes in use
    ng=size(gridunc)/maxageclass*nageclass
    allocate(g(ng))
    g=reshape(real(gridunc,8), [ng])
    call put_d('gridunc', g, ng)
    deallocate(g)
    ng=size(drygridunc)/maxageclass*nageclass
    allocate(g(ng))
    g=reshape(real(drygridunc,8), [ng])
    call put_d('drygridunc', g, ng)
    g=reshape(real(wetgridunc,8), [ng])
    call put_d('wetgridunc', g, ng)
    deallocate(g)
    if (do_concn .eq. 1) then
      ng=size(griduncn)/maxageclass*nageclass
      allocate(g(ng))
      g=reshape(real(griduncn,8), [ng])
      call put_d('griduncn', g, ng)
      deallocate(g)
      ng=size(drygriduncn)/maxageclass*nageclass
      allocate(g(ng))
      g=reshape(real(drygriduncn,8), [ng])
      call put_d('drygriduncn', g, ng)
      g=reshape(real(wetgriduncn,8), [ng])
      call put_d('wetgriduncn', g, ng)
      deallocate(g)
    end if
    if (numreceptor .gt. 0) then
      ng=numreceptor*nspec
      allocate(g(ng))
      g=reshape(real(creceptor(1:numreceptor,1:nspec),8), [ng])
      call put_d('creceptor', g, ng)
      deallocate(g)
    end if
  end subroutine dump_grids

  subroutine dump_state()
    integer :: np, kk
    integer, allocatable :: it(:)
    np=numpart
    allocate(it(np))
    tmp(1:np)=xtra1(1:np); call put_d('xtra1', tmp, np)
    tmp(1:np)=ytra1(1:np); call put_d('ytra1', tmp, np)
    tmp(1:np)=ztra1(1:np); call put_d('ztra1', tmp, np)
    tmp(1:np)=uap(1:np);   call put_d('uap', tmp, np)
    tmp(1:np)=ucp(1:np);   call put_d('ucp', tmp, np)
    tmp(1:np)=uzp(1:np);   call put_d('uzp', tmp, np)
    tmp(1:np)=us(1:np);    call put_d('us', tmp, np)
    tmp(1:np)=vs(1:np);    call put_d('vs', tmp, np)
    tmp(1:np)=ws(1:np);    call put_d('ws', tmp, np)
    it(1:np)=idt(1:np);    call put_i('idt', it, np)
    it(1:np)=itra1(1:np);  call put_i('itra1', it, np)
    do kk=1,np
      it(kk)=int(cbt(kk))
    end do
    call put_i('cbt', it, np)
    do kk=1,nspec
      tmp(1:np)=xmass1(1:np,kk); call put_d('xmass1', tmp, np)
    end do
    if (DRYBKDEP .or. WETBKDEP) then
      do kk=1,nspec
        tmp(1:np)=xscav_frac1(1:np,kk); call put_d('xscav_frac1', tmp, np)
      end do
    end if
    deallocate(it)
  end subroutine dump_state

end program flexref
