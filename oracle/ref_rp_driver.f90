! TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
!
! ref_rp_driver: drives the *unmodified* reference routine readpartpositions
! (/root/reference/src/readpartpositions.f90, the warm start from a particle dump; SURVEY section 8 f4)
! on a directory that holds a dump `partposit_end`; writes the `header` file the routine insists on
! reading first (our own minimal writer: the record sequence readpartpositions.f90:59-113 skips through),
! calls the routine and dumps the particle arrays it filled.
! This file is our own code: it contains no reference source.
!
! Usage:  rpref_rK scenario.bin dir/ out.bin [gpu]     (dir/ holds partposit_end; header is written there)
!   gpu: flexgpu_readpartpositions (flexpart_amd/fortran/flexgpu_mod.f90) parses the dump on the MI355X and the
!        particle arrays come back through flexgpu_download_particles (needs a GPU).
! Record format as oracle/ref_driver.f90.

program rpref
  use par_mod
  use com_mod
  use point_mod
  use flexgpu_mod
  implicit none
  integer :: use_gpu, gerr
  character(len=256) :: gmsg
  character(len=16) :: arg4
  integer, parameter :: uin=31, uout=32, uh=33
  character(len=512) :: fscen, fdir, fout
  character(len=16) :: name
  integer(kind=4) :: dtype
  integer(kind=8) :: cnt
  integer, allocatable :: ibuf(:)
  real(kind=8), allocatable :: dbuf(:), t(:)
  integer :: ios, n, i, j, ix, ibdatein, ibtimein, maxp
  real(kind=dp) :: juldate

  call get_command_argument(1, fscen)
  call get_command_argument(2, fdir)
  call get_command_argument(3, fout)
  use_gpu = 0
  if (command_argument_count() .ge. 4) then
    call get_command_argument(4, arg4)
    if (trim(arg4) .eq. 'gpu') use_gpu = 1
  end if
  path(2) = trim(fdir); length(2) = len_trim(fdir)
  nspec=1; ldirect=1; mintime=1; itsplit=999999999; numpoint=1; numxgrid=1
  ibdate=20200101; ibtime=0; ibdatein=20200101; ibtimein=0; maxp=1000
  dx=1.; dy=1.; xlon0=0.; ylat0=0.

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
    case ('geom');     dx=dbuf(1); dy=dbuf(2); xlon0=dbuf(3); ylat0=dbuf(4)
    case ('nspec');    nspec=ibuf(1)
    case ('restart')   ! ibdate ibtime ibdatein ibtimein ldirect mintime itsplit (unused: nclassunc is a par_mod constant) numpoint maxpart
      ibdate=ibuf(1); ibtime=ibuf(2); ibdatein=ibuf(3); ibtimein=ibuf(4); ldirect=ibuf(5)
      mintime=ibuf(6); itsplit=ibuf(7); numpoint=ibuf(9); maxp=ibuf(10)
    case default
      write(*,*) 'ref_rp_driver: unknown record ', trim(name); stop 1
    end select
  end do
  close(uin)
  bdate = juldate(ibdate, ibtime)
  do i=1,nspec
    species(i) = 'SPEC001'
  end do
  ipout=0; iflux=0; linit_cond=0
  call com_mod_allocate_part(maxp)

  ! the header file: only the records readpartpositions reads or skips (readpartpositions.f90:59-113)
  open(uh, file=trim(fdir)//'header', form='unformatted', status='replace')
  write(uh) ibdatein, ibtimein
  write(uh) 0
  write(uh) 0
  write(uh) 0
  write(uh) 0
  write(uh) 3*nspec
  do i=1,nspec
    write(uh) 0
    write(uh) 0
    write(uh) 1, species(i)(1:7)
  end do
  write(uh) numpoint
  do i=1,numpoint
    write(uh) 0
    write(uh) 0
    write(uh) 0
    write(uh) 0
    do j=1,nspec
      write(uh) 0
      write(uh) 0
      write(uh) 0
    end do
  end do
  write(uh) 0
  write(uh) 0
  do ix=0,numxgrid-1
    write(uh) 0
  end do
  close(uh)

  if (use_gpu .eq. 1) then
    nx=10; ny=10; nz=3; nxmin1=9; nymin1=9; nmixz=2
    ldirect=1; lsynctime=900; method=1; ctl=0.2; ifine=4; turbswitch=.true.; cblflag=0
    mdomainfill=0; lsettling=.false.; DRYDEP=.false.; nageclass=1; lage(1)=999999999
    xglobal=.false.; nglobal=.false.; sglobal=.false.; switchnorthg=999999.; switchsouthg=999999.
    allocate(xmass(1,maxspec), npart(1)); xmass=1.; npart(1)=1
    call flexgpu_init(gerr, nmaxpart=maxp, defer_height=.true.)
    if (gerr .eq. 0) call flexgpu_readpartpositions(ibdatein, ibtimein, gerr)
    if (gerr .eq. 0 .and. numpart .gt. 0) call flexgpu_download_particles(1, numpart, gerr)
    if (gerr .ne. 0) then
      call flexgpu_last_error(gmsg); write(*,*) 'flexgpu: ', trim(gmsg); stop 1
    end if
  else
    call readpartpositions
  end if

  open(uout, file=trim(fout), access='stream', form='unformatted', status='replace')
  call put_i('numpart', (/numpart, numparticlecount/), 2)
  allocate(t(max(numpart,1)))
  t(1:numpart)=xtra1(1:numpart); call put_d('xtra1', t, numpart)
  t(1:numpart)=ytra1(1:numpart); call put_d('ytra1', t, numpart)
  t(1:numpart)=ztra1(1:numpart); call put_d('ztra1', t, numpart)
  call put_i('npoint', npoint(1:numpart), numpart)
  call put_i('itramem', itramem(1:numpart), numpart)
  call put_i('nclass', nclass(1:numpart), numpart)
  call put_i('idt', idt(1:numpart), numpart)
  call put_i('itra1', itra1(1:numpart), numpart)
  call put_i('itrasplit', itrasplit(1:numpart), numpart)
  do j=1,nspec
    t(1:numpart)=xmass1(1:numpart,j); call put_d('xmass1', t, numpart)
  end do
  write(uout) 'END             ', 1_4, 0_8
  close(uout)
contains
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
end program rpref
