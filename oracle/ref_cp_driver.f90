! TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
!
! ref_cp_driver: calls the *unmodified* leaf routines of the reference's calcpar that compile in this image --
! scalev (scalev.f90), ew (ew.f90), f_qvsat (qvsat.f90), compiled where they lie by oracle/build_ref.sh -- on arrays
! of arguments, so that oracle/calcpar_oracle.c can be pinned against them bit for bit (SURVEY section 8 f1).
! calcpar.f90, obukhov.f90 and richardson.f90 themselves `use class_gribfile` (ecCodes) and cannot be built here.
! This file is our own code: it contains no reference source.
!
! Usage:  cpref_rK in.bin out.bin      in: n (i4), then ps, t, td, stress (n f64 each); out: scalev, ew(td), f_qvsat(ps,t)
program cpref
  implicit none
  character(len=512) :: fin, fout
  integer(kind=4) :: n
  integer :: i
  real(kind=8), allocatable :: a(:,:), o(:,:)
  real :: scalev, ew, f_qvsat
  real :: ps, t, td, st
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(31, file=trim(fin), access='stream', form='unformatted', status='old')
  read(31) n
  allocate(a(n,4), o(n,3))
  read(31) a
  close(31)
  do i=1,n
    ps=a(i,1); t=a(i,2); td=a(i,3); st=a(i,4)
    o(i,1)=scalev(ps,t,td,st)
    o(i,2)=ew(td)
    o(i,3)=f_qvsat(ps,t)
  end do
  open(32, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(32) o
  close(32)
end program cpref
