! TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
!
! ref_rel_driver: drives the *unmodified* reference routine releaseparticles
! (/root/reference/src/releaseparticles.f90, compiled where it lies by oracle/build_ref.sh) on release
! points, fields and (optionally) particles that already exist, for a list of times; after every call
! it runs the particle-splitting block of the time manager (our restatement of timemanager.f90:473-504,
! which is inline code there) and dumps the particle arrays (SURVEY section 8 f2).
! This file is our own code: it contains no reference source, only calls into it and assignments to
! its module variables.
!
! Usage:  relref_rK scenario.bin out.bin [gpu]
!   gpu: the same com_mod / point_mod arrays go to the MI355X engine through
!        flexpart_amd/fortran/flexgpu_mod.f90; flexgpu_releaseparticles and flexgpu_split_particles
!        replace the Fortran routine and the splitting block (needs a GPU).
! Record format as oracle/ref_driver.f90: {name*16, dtype i4 (1=i32, 2=f64), count i8, payload}.

program relref
  use par_mod
  use com_mod
  use point_mod
  use xmass_mod
  use flexgpu_mod
  implicit none

  integer, parameter :: uin=31, uout=32
  integer :: use_gpu, gerr
  character(len=256) :: gmsg
  character(len=16) :: arg3
  character(len=512) :: fscen, fout
  character(len=16) :: name
  integer(kind=4) :: dtype
  integer(kind=8) :: cnt
  integer, allocatable :: ibuf(:), times(:)
  real(kind=8), allocatable :: dbuf(:), tmp(:)
  integer :: ios, n, gnx, gny, gnz, np, ks, i, j, k, it, ntimes, itime, maxp, do_split, ii1, jj1
  real(kind=dp) :: juldate

  call get_command_argument(1, fscen)
  call get_command_argument(2, fout)
  use_gpu = 0
  if (command_argument_count() .ge. 3) then
    call get_command_argument(3, arg3)
    if (trim(arg3) .eq. 'gpu') use_gpu = 1
  end if
  ! defaults (readcommand.f90 / readreleases.f90 state)
  nspec = 1; numpart = 0; numparticlecount = 0; numpoint = 1
  ldirect = 1; lsynctime = 900; mintime = 1; itsplit = 999999999; ind_rel = 0; mquasilag = 0
  xglobal = .false.; numbnests = 0
  DRYBKDEP = .false.; WETBKDEP = .false.
  area_hour = 1.; point_hour = 1.; area_dow = 1.; point_dow = 1.
  ibdate = 20200101; ibtime = 0
  memind(1) = 1; memind(2) = 2
  gnx = 0; gny = 0; gnz = 0; np = 0; ntimes = 0; maxp = 0; do_split = 0

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
    case ('grid')
      gnx=ibuf(1); gny=ibuf(2); gnz=ibuf(3)
      if (gnx.gt.nxmax .or. gny.gt.nymax .or. gnz.gt.nzmax) stop 'grid too large'
      nx=gnx; ny=gny; nz=gnz; nxmin1=nx-1; nymin1=ny-1
    case ('geom');     dx=dbuf(1); dy=dbuf(2); xlon0=dbuf(3); ylat0=dbuf(4)
    case ('xglobal');  xglobal=(ibuf(1).ne.0)
    case ('height');   height(1:n)=dbuf(1:n)
    case ('nspec');    nspec=ibuf(1)
    case ('bdate');    ibdate=ibuf(1); ibtime=ibuf(2)
    case ('switches')  ! ldirect lsynctime mintime itsplit ind_rel mquasilag maxpart do_split
      ldirect=ibuf(1); lsynctime=ibuf(2); mintime=ibuf(3); itsplit=ibuf(4); ind_rel=ibuf(5); mquasilag=ibuf(6)
      maxp=ibuf(7); do_split=ibuf(8)
    case ('times');    ntimes=n; allocate(times(n)); times=ibuf(1:n)
    case ('oro');      call get2(oro)
    case ('rho2');     call get3(rho(:,:,:,2))      ! releaseparticles reads the literal slot 2 (:233-257,:315-325)
    case ('tt2');      call get3(tt(:,:,:,2))
#ifdef FLEXREF_NESTS
    ! --- one nested wind field (releaseparticles.f90:196-226,231-341) ------------------------------
    case ('nest')
      numbnests=1; nxn(1)=ibuf(1); nyn(1)=ibuf(2)
      if (nxn(1).gt.nxmaxn .or. nyn(1).gt.nymaxn) stop 'nest too large'
      call com_mod_allocate_nests
      uun=0.; vvn=0.; wwn=0.; ttn=0.; rhon=0.; drhodzn=0.
    case ('nestcorners')
      xln(1)=dbuf(1); yln(1)=dbuf(2); xrn(1)=dbuf(3); yrn(1)=dbuf(4); xresoln(1)=dbuf(5); yresoln(1)=dbuf(6)
      xresoln(0)=1.; yresoln(0)=1.
    case ('oron')
      do jj1=0,nyn(1)-1
        do ii1=0,nxn(1)-1
          oron(ii1,jj1,1)=dbuf(1+ii1+nxn(1)*jj1)
        end do
      end do
    case ('rhon2'); call get3n(rhon(:,:,:,2,1))
    case ('ttn2');  call get3n(ttn(:,:,:,2,1))
    case ('par_nxmax')
      if (ibuf(1).ne.nxmax) stop 'the scenario par_nxmax differs from par_mod nxmax'
#else
    case ('nest', 'nestcorners', 'oron', 'rhon2', 'ttn2', 'par_nxmax')
      stop 'this build has no nests (par_mod maxnests = 0)'
#endif
    ! --- release points: point_mod arrays as readreleases.f90 allocates them ---------------------
    case ('numpoint')
      numpoint=ibuf(1)
      allocate(ireleasestart(numpoint), ireleaseend(numpoint), npart(numpoint), kindz(numpoint))
      allocate(xpoint1(numpoint), xpoint2(numpoint), ypoint1(numpoint), ypoint2(numpoint))
      allocate(zpoint1(numpoint), zpoint2(numpoint), xmass(numpoint,maxspec), rho_rel(numpoint), xmasssave(numpoint))
      xmass=0.; rho_rel=0.; xmasssave=0.
    case ('ireleasestart'); ireleasestart(1:n)=ibuf(1:n)
    case ('ireleaseend');   ireleaseend(1:n)=ibuf(1:n)
    case ('npart_rel');     npart(1:n)=ibuf(1:n)
    case ('kindz');         do i=1,n; kindz(i)=int(ibuf(i),2); end do
    case ('xpoint1');  xpoint1(1:n)=dbuf(1:n)
    case ('xpoint2');  xpoint2(1:n)=dbuf(1:n)
    case ('ypoint1');  ypoint1(1:n)=dbuf(1:n)
    case ('ypoint2');  ypoint2(1:n)=dbuf(1:n)
    case ('zpoint1');  zpoint1(1:n)=dbuf(1:n)
    case ('zpoint2');  zpoint2(1:n)=dbuf(1:n)
    case ('xmass')     ! (numpoint, nspec), species-major
      do ks=1,nspec
        xmass(1:numpoint,ks)=dbuf(1+(ks-1)*numpoint:ks*numpoint)
      end do
    case ('point_hour'); do k=1,24; point_hour(1:nspec,k)=dbuf(1+(k-1)*nspec:k*nspec); end do
    case ('area_hour');  do k=1,24; area_hour(1:nspec,k)=dbuf(1+(k-1)*nspec:k*nspec); end do
    case ('point_dow');  do k=1,7; point_dow(1:nspec,k)=dbuf(1+(k-1)*nspec:k*nspec); end do
    case ('area_dow');   do k=1,7; area_dow(1:nspec,k)=dbuf(1+(k-1)*nspec:k*nspec); end do
    ! --- particles that exist before the first call (a run in progress) ----------------------------
    case ('npart')
      np=ibuf(1)
      ! maxpart is a compile-time size of the reference (par_mod.f90:210): the scenario must use the same value
      if (maxp .ne. maxpart) stop 'the scenario maxpart differs from par_mod maxpart'
      call com_mod_allocate_part(maxpart)
      itra1(:)=-999999999; npoint(:)=0; nclass(:)=0; idt(:)=0; itramem(:)=0; itrasplit(:)=999999999
      xtra1(:)=0.; ytra1(:)=0.; ztra1(:)=0.; xmass1(:,:)=0.
      uap(:)=0.; ucp(:)=0.; uzp(:)=0.; us(:)=0.; vs(:)=0.; ws(:)=0.; cbt(:)=1
      numpart=np
    case ('xtra1');    xtra1(1:n)=dbuf(1:n)
    case ('ytra1');    ytra1(1:n)=dbuf(1:n)
    case ('ztra1');    ztra1(1:n)=dbuf(1:n)
    case ('itra1');    itra1(1:n)=ibuf(1:n)
    case ('itramem');  itramem(1:n)=ibuf(1:n)
    case ('itrasplit'); itrasplit(1:n)=ibuf(1:n)
    case ('npoint');   npoint(1:n)=ibuf(1:n)
    case ('nclass');   nclass(1:n)=ibuf(1:n)
    case ('idt');      idt(1:n)=ibuf(1:n)
    case ('uap');      uap(1:n)=dbuf(1:n)
    case ('xmass1')
      do ks=1,nspec
        xmass1(1:np,ks)=dbuf(1+(ks-1)*np:ks*np)
      end do
    case default
      write(*,*) 'ref_rel_driver: unknown record ', trim(name)
      stop 1
    end select
  end do
  close(uin)
  bdate = juldate(ibdate, ibtime)

  open(uout, file=trim(fout), access='stream', form='unformatted', status='replace')
  allocate(tmp(max(maxpart, numpoint)))

  if (use_gpu .eq. 1) then
    ! run switches flexgpu_init reads (no trajectory step is taken here)
    method=1; ctl=0.2; ifine=4; turbswitch=.true.; cblflag=0
    ipout=0; iflux=0; linit_cond=0
    mdomainfill=0; lsettling=.false.; DRYDEP=.false.; nageclass=1; lage(1)=999999999
    nglobal=.false.; sglobal=.false.; switchnorthg=999999.; switchsouthg=999999.
    dxconst=180./(dx*r_earth*pi); dyconst=180./(dy*r_earth*pi)
    nmixz=nz
    do i=2,nz
      if (height(i).gt.hmixmax) then
        nmixz=i; exit
      end if
    end do
    rho(:,:,:,1)=rho(:,:,:,2); tt(:,:,:,1)=tt(:,:,:,2)
    ustar=0.3; wstar=1.; oli=0.01; hmix=500.; tropopause=10000.
    memtime(1)=0; memtime(2)=10800; lwindinterv=10800
    call flexgpu_init(gerr, nmaxpart=maxpart)
    if (gerr .eq. 0) call flexgpu_use_table_rng(gerr)          ! parity mode: the serial ran1 stream is replayed
    if (gerr .eq. 0) call flexgpu_upload_fields(1, gerr)
    if (gerr .eq. 0) call flexgpu_upload_fields(2, gerr)
    if (gerr .eq. 0) call flexgpu_set_windtime(gerr)
    if (gerr .eq. 0) call flexgpu_upload_diag_fields(0, gerr)   ! oro
    if (gerr .eq. 0) call flexgpu_upload_diag_fields(2, gerr)   ! tt of slot 2 (kindz = 3)
#ifdef FLEXREF_NESTS
    if (numbnests .gt. 0) then
      rhon(:,:,:,1,:)=rhon(:,:,:,2,:); ttn(:,:,:,1,:)=ttn(:,:,:,2,:)
      hmixn=500.; ustarn=0.3; wstarn=1.; olin=0.01; tropopausen=10000.; vdepn=0.
      if (gerr .eq. 0) call flexgpu_upload_nests(gerr)
      if (gerr .eq. 0) call flexgpu_upload_diag_nest_fields(gerr)
    end if
#endif
    if (gerr .eq. 0 .and. numpart .gt. 0) call flexgpu_upload_particles(1, numpart, gerr)
    if (gerr .eq. 0) call flexgpu_release_init(gerr)
    if (gerr .ne. 0) call gpu_fail('set-up')
  end if

  do it=1,ntimes
    itime=times(it)
    if (use_gpu .eq. 1) then
      call flexgpu_releaseparticles(itime, gerr)
      if (gerr .ne. 0) call gpu_fail('flexgpu_releaseparticles')
      if (do_split .eq. 1) then
        call flexgpu_split_particles(itime, gerr)
        if (gerr .ne. 0) call gpu_fail('flexgpu_split_particles')
      end if
      call flexgpu_download_particles(1, numpart, gerr)
      if (gerr .ne. 0) call gpu_fail('flexgpu_download_particles')
    else
      call releaseparticles(itime)
      if (do_split .eq. 1) call split_block(itime)
    end if
    call dump(itime)
  end do
  name='END'
  write(uout) name, 1_4, 0_8
  close(uout)
  if (use_gpu .eq. 1) call flexgpu_finalize()

contains

  subroutine gpu_fail(where)
    character(len=*), intent(in) :: where
    call flexgpu_last_error(gmsg)
    write(*,*) 'ref_rel_driver: ', where, ' failed: ', gerr, ' ', trim(gmsg)
    stop 2
  end subroutine gpu_fail

  ! our restatement of the splitting block, inline code of the time manager (timemanager.f90:473-504)
  subroutine split_block(itime)
    integer, intent(in) :: itime
    integer :: jj, nn, kk
    if (ldirect*itime.ge.ldirect*itsplit) then
      nn=numpart
      do jj=1,numpart
        if (ldirect*itime.ge.ldirect*itrasplit(jj)) then
          if (nn.lt.maxpart) then
            nn=nn+1
            itrasplit(jj)=2*(itrasplit(jj)-itramem(jj))+itramem(jj)
            itrasplit(nn)=itrasplit(jj)
            itramem(nn)=itramem(jj)
            itra1(nn)=itra1(jj)
            idt(nn)=idt(jj)
            npoint(nn)=npoint(jj)
            nclass(nn)=nclass(jj)
            xtra1(nn)=xtra1(jj)
            ytra1(nn)=ytra1(jj)
            ztra1(nn)=ztra1(jj)
            uap(nn)=uap(jj)
            ucp(nn)=ucp(jj)
            uzp(nn)=uzp(jj)
            us(nn)=us(jj)
            vs(nn)=vs(jj)
            ws(nn)=ws(jj)
            cbt(nn)=cbt(jj)
            do kk=1,nspec
              xmass1(jj,kk)=xmass1(jj,kk)/2.
              xmass1(nn,kk)=xmass1(jj,kk)
            end do
          endif
        endif
      end do
      numpart=nn
    endif
  end subroutine split_block

  subroutine dump(itime)
    integer, intent(in) :: itime
    integer :: m, kk
    integer, allocatable :: iv(:)
    m=numpart
    allocate(iv(max(m,3)))
    iv(1)=itime; iv(2)=numpart; iv(3)=numparticlecount
    call put_i('state', iv, 3)
    tmp(1:m)=xtra1(1:m); call put_d('xtra1', tmp, m)
    tmp(1:m)=ytra1(1:m); call put_d('ytra1', tmp, m)
    tmp(1:m)=ztra1(1:m); call put_d('ztra1', tmp, m)
    tmp(1:m)=uap(1:m);   call put_d('uap', tmp, m)
    iv(1:m)=itra1(1:m);     call put_i('itra1', iv, m)
    iv(1:m)=itramem(1:m);   call put_i('itramem', iv, m)
    iv(1:m)=itrasplit(1:m); call put_i('itrasplit', iv, m)
    iv(1:m)=idt(1:m);       call put_i('idt', iv, m)
    iv(1:m)=npoint(1:m);    call put_i('npoint', iv, m)
    iv(1:m)=nclass(1:m);    call put_i('nclass', iv, m)
    do kk=1,nspec
      tmp(1:m)=xmass1(1:m,kk); call put_d('xmass1', tmp, m)
    end do
    if (use_gpu .eq. 0) then
      tmp(1:numpoint)=xmasssave(1:numpoint); call put_d('xmasssave', tmp, numpoint)
      tmp(1:numpoint)=rho_rel(1:numpoint);   call put_d('rho_rel', tmp, numpoint)
    end if
    deallocate(iv)
  end subroutine dump

  subroutine put_i(nm0, a, m)
    character(len=*), intent(in) :: nm0
    integer, intent(in) :: m
    integer, intent(in) :: a(m)
    character(len=16) :: nm
    nm = nm0
    write(uout) nm, 1_4, int(m,8), a(1:m)
  end subroutine put_i
  subroutine put_d(nm0, a, m)
    character(len=*), intent(in) :: nm0
    integer, intent(in) :: m
    real(kind=8), intent(in) :: a(m)
    character(len=16) :: nm
    nm = nm0
    write(uout) nm, 2_4, int(m,8), a(1:m)
  end subroutine put_d

  subroutine get2(f)
    real, intent(inout) :: f(0:nxmax-1,0:nymax-1)
    integer :: ii,jj
    do jj=0,gny-1
      do ii=0,gnx-1
        f(ii,jj)=dbuf(1+ii+gnx*jj)
      end do
    end do
  end subroutine get2

#ifdef FLEXREF_NESTS
  subroutine get3n(f)
    real, intent(inout) :: f(0:nxmaxn-1,0:nymaxn-1,nzmax)
    integer :: ii,jj,kk
    do kk=1,gnz
      do jj=0,nyn(1)-1
        do ii=0,nxn(1)-1
          f(ii,jj,kk)=dbuf(1+ii+nxn(1)*(jj+nyn(1)*(kk-1)))
        end do
      end do
    end do
  end subroutine get3n
#endif

  subroutine get3(f)
    real, intent(inout) :: f(0:nxmax-1,0:nymax-1,nzmax)
    integer :: ii,jj,kk
    do kk=1,gnz
      do jj=0,gny-1
        do ii=0,gnx-1
          f(ii,jj,kk)=dbuf(1+ii+gnx*(jj+gny*(kk-1)))
        end do
      end do
    end do
  end subroutine get3

end program relref
