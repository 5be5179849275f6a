! TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
!
! ref_po_driver: drives the *unmodified* reference routine partoutput
! (/root/reference/src/partoutput.f90, compiled where it lies by oracle/build_ref.sh) on fields
! and particles read from a scenario file; the routine itself writes the dump
! <outdir>/partposit_end (SURVEY section 8 f4: the on-disk particle format).
! This file is our own code: it contains no reference source, only calls into it and
! assignments to its module variables.
!
! Usage:  poref_rK scenario.bin outdir/ [gpu]
!   gpu: the same com_mod arrays go to the MI355X engine through flexpart_amd/fortran/flexgpu_mod.f90 and
!        flexgpu_partoutput writes the file instead of the Fortran routine (needs a GPU).
! Record format as oracle/ref_driver.f90: {name*16, dtype i4 (1=i32, 2=f64), count i8, payload};
! fields travel compact, (nx,ny[,nz],slot) x fastest, as f64.

program poref
  use par_mod
  use com_mod
  use point_mod
  use flexgpu_mod
  implicit none

  integer :: use_gpu, gerr
  character(len=256) :: gmsg
  character(len=16) :: arg3
  integer, parameter :: uin=31
  character(len=512) :: fscen, fout
  character(len=16) :: name
  integer(kind=4) :: dtype
  integer(kind=8) :: cnt
  integer, allocatable :: ibuf(:)
  real(kind=8), allocatable :: dbuf(:)
  integer :: ios, n, gnx, gny, gnz, np, itime_out, ks, i
  real(kind=dp) :: juldate

  call get_command_argument(1, fscen)
  call get_command_argument(2, fout)
  use_gpu = 0
  if (command_argument_count() .ge. 3) then
    call get_command_argument(3, arg3)
    if (trim(arg3) .eq. 'gpu') use_gpu = 1
  end if
  path(2) = trim(fout)
  length(2) = len_trim(fout)
  ipout = 2                      ! -> file name partposit_end (partoutput.f90:84-86)
  bdate = juldate(20200101, 0)
  nspec = 1; numpart = 0; itime_out = 0
  memind(1) = 1; memind(2) = 2
  gnx = 0; gny = 0; gnz = 0; np = 0

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
    case ('height');   height(1:n)=dbuf(1:n)
    case ('memtime');  memtime(1)=ibuf(1); memtime(2)=ibuf(2)
    case ('memind');   memind(1)=ibuf(1); memind(2)=ibuf(2)
    case ('nspec');    nspec=ibuf(1)
    case ('itime');    itime_out=ibuf(1)
    case ('oro');      call get2(oro)
    case ('pv');       call get3(pv(:,:,:,1), 0); call get3(pv(:,:,:,2), 1)
    case ('qv');       call get3(qv(:,:,:,1), 0); call get3(qv(:,:,:,2), 1)
    case ('tt');       call get3(tt(:,:,:,1), 0); call get3(tt(:,:,:,2), 1)
    case ('rho');      call get3(rho(:,:,:,1), 0); call get3(rho(:,:,:,2), 1)
    case ('hmix');     call get2s(hmix(:,:,1,1), 0); call get2s(hmix(:,:,1,2), 1)
    case ('tropopause'); call get2s(tropopause(:,:,1,1), 0); call get2s(tropopause(:,:,1,2), 1)
    case ('npart')
      np=ibuf(1)
      call com_mod_allocate_part(np)
      numpart=np
    case ('xtra1');    xtra1(1:np)=dbuf(1:np)
    case ('ytra1');    ytra1(1:np)=dbuf(1:np)
    case ('ztra1');    ztra1(1:np)=dbuf(1:np)
    case ('itra1');    itra1(1:np)=ibuf(1:np)
    case ('itramem');  itramem(1:np)=ibuf(1:np)
    case ('npoint');   npoint(1:np)=ibuf(1:np)
    case ('xmass1')
      do ks=1,nspec
        xmass1(1:np,ks)=dbuf(1+(ks-1)*np:ks*np)
      end do
    case default
      write(*,*) 'ref_po_driver: unknown record ', trim(name)
      stop 1
    end select
  end do
  close(uin)

  if (use_gpu .eq. 1) then
    ! run switches flexgpu_init reads (no particle step is taken here)
    ldirect=1; lsynctime=900; method=1; mintime=1; ctl=0.2; ifine=4; turbswitch=.true.; cblflag=0
    iflux=0; linit_cond=0          ! (ipout = 2 above: the dump at the end of the run, partoutput.f90:84-86)
    mdomainfill=0; lsettling=.false.; DRYDEP=.false.; nageclass=1; lage(1)=999999999
    xglobal=.false.; nglobal=.false.; sglobal=.false.; switchnorthg=999999.; switchsouthg=999999.
    dxconst=180./(dx*r_earth*pi); dyconst=180./(dy*r_earth*pi)
    do i=2,nz
      if (height(i).gt.hmixmax) then
        nmixz=i; exit
      end if
    end do
    numpoint=1; allocate(xmass(1,maxspec), npart(1)); xmass=1.; npart(1)=max(np,1)
    ustar=0.3; wstar=1.; oli=0.01
    call flexgpu_init(gerr, nmaxpart=max(np,1))
    if (gerr .eq. 0) call flexgpu_upload_fields(1, gerr)
    if (gerr .eq. 0) call flexgpu_upload_fields(2, gerr)
    if (gerr .eq. 0) call flexgpu_set_windtime(gerr)
    if (gerr .eq. 0) call flexgpu_upload_diag_fields(0, gerr)
    if (gerr .eq. 0) call flexgpu_upload_diag_fields(1, gerr)
    if (gerr .eq. 0) call flexgpu_upload_diag_fields(2, gerr)
    if (gerr .eq. 0 .and. np .gt. 0) call flexgpu_upload_particles(1, np, gerr)
    if (gerr .eq. 0) call flexgpu_partoutput(itime_out, gerr)
    if (gerr .ne. 0) then
      call flexgpu_last_error(gmsg); write(*,*) 'flexgpu: ', trim(gmsg); stop 1
    end if
  else
    call partoutput(itime_out)
  end if

contains
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
  subroutine get2s(a, m)
    real, intent(inout) :: a(0:nxmax-1,0:nymax-1)
    integer, intent(in) :: m
    integer :: ix, jy
    a=0.
    do jy=0,gny-1
      do ix=0,gnx-1
        a(ix,jy)=dbuf(1+ix+gnx*(jy+gny*m))
      end do
    end do
  end subroutine get2s
  subroutine get3(a, m)
    real, intent(inout) :: a(0:nxmax-1,0:nymax-1,nzmax)
    integer, intent(in) :: m
    integer :: ix, jy, k
    a=0.
    do k=1,gnz
      do jy=0,gny-1
        do ix=0,gnx-1
          a(ix,jy,k)=dbuf(1+ix+gnx*(jy+gny*((k-1)+gnz*m)))
        end do
      end do
    end do
  end subroutine get3
end program poref
