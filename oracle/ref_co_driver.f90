! TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
!
! ref_co_driver: drives the *unmodified* reference routine concoutput
! (/root/reference/src/concoutput.f90, compiled where it lies by oracle/build_ref.sh; SURVEY section 8 f4:
! the sparse grid_conc_* writer) on output grids read from a scenario file.  The routine itself writes
! <outdir>/grid_conc_<date><time>_<species> (and its side files dates, factor_drygrid).
! This file is our own code: it contains no reference source, only calls into it, assignments to its
! module variables and the allocations outgrid_init.f90:192-290 would make.
!
! Usage:  coref_rK scenario.bin outdir/ [nest]
!   nest: the grids go to the nested output grid's arrays (griduncn, arean, volumen ...) and concoutput_nest writes
!         grid_conc_nest_<date><time>_<species> (concoutput_nest.f90).
! Record format as oracle/ref_driver.f90; grids travel compact, x fastest, as f64.

program coref
  use par_mod
  use com_mod
  use unc_mod
  use outg_mod
  use point_mod
  implicit none
  integer, parameter :: uin=31
  character(len=512) :: fscen, fout
  character(len=16) :: name
  integer(kind=4) :: dtype
  integer(kind=8) :: cnt
  integer, allocatable :: ibuf(:)
  real(kind=8), allocatable :: dbuf(:)
  integer :: ios, n, itime_out, nxg, nyg, nzg, i, ix, jy, kz, ks, use_nest, kc, ncls = 1
  character(len=16) :: arg3
  real :: outnum, gtu
  real(dep_prec) :: wtu, dtu
  real(kind=dp) :: juldate

  call get_command_argument(1, fscen)
  call get_command_argument(2, fout)
  use_nest = 0
  if (command_argument_count() .ge. 3) then
    call get_command_argument(3, arg3)
    if (trim(arg3) .eq. 'nest') use_nest = 1
  end if
  path(2) = trim(fout); length(2) = len_trim(fout)
  bdate = juldate(20200101, 0)
  ldirect=1; iout=1; nspec=1; maxpointspec_act=1; nageclass=1; numreceptor=0
  WETDEP=.false.; DRYDEP=.false.; DRYBKDEP=.false.; WETBKDEP=.false.
  memind(1)=1; memind(2)=2; loutaver=3600; outnum=4.; itime_out=3600
  nx=10; ny=10; nz=3; nxmin1=9; nymin1=9; dx=1.; dy=1.; xlon0=0.; ylat0=0.
  height(1)=0.; height(2)=1000.; height(3)=50000.
  rho=1.2; rho_dry=1.0
  allocate(xmass(1,maxspec)); xmass=1.

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
    case ('outgrid')   ! numxgrid numygrid numzgrid nspec wetdep drydep itime
      nxg=ibuf(1); nyg=ibuf(2); nzg=ibuf(3); nspec=ibuf(4)
      WETDEP=(ibuf(5).ne.0); DRYDEP=(ibuf(6).ne.0); itime_out=ibuf(7)
      numxgrid=nxg; numygrid=nyg; numzgrid=nzg
      allocate(outheight(nzg), area(0:nxg-1,0:nyg-1), volume(0:nxg-1,0:nyg-1,nzg))
      allocate(grid(0:nxg-1,0:nyg-1,nzg), gridsigma(0:nxg-1,0:nyg-1,nzg), densityoutgrid(0:nxg-1,0:nyg-1,nzg))
      allocate(densitydrygrid(0:nxg-1,0:nyg-1,nzg), factor_drygrid(0:nxg-1,0:nyg-1,nzg), factor3d(0:nxg-1,0:nyg-1,nzg))
      allocate(sparse_dump_r(nxg*nyg*nzg), sparse_dump_u(nxg*nyg*nzg), sparse_dump_i(nxg*nyg*nzg))
      allocate(wetgrid(0:nxg-1,0:nyg-1), drygrid(0:nxg-1,0:nyg-1), wetgridsigma(0:nxg-1,0:nyg-1), drygridsigma(0:nxg-1,0:nyg-1))
      allocate(gridunc(0:nxg-1,0:nyg-1,nzg,maxspec,maxpointspec_act,nclassunc,maxageclass))
      allocate(wetgridunc(0:nxg-1,0:nyg-1,maxspec,maxpointspec_act,nclassunc,maxageclass))
      allocate(drygridunc(0:nxg-1,0:nyg-1,maxspec,maxpointspec_act,nclassunc,maxageclass))
      gridunc=0.; wetgridunc=0.; drygridunc=0.
      if (use_nest .eq. 1) then      ! the nested output grid shares the work arrays (outgrid_init.f90:245-290 sizes them by max)
        numxgridn=nxg; numygridn=nyg; nested_output=1
        allocate(arean(0:nxg-1,0:nyg-1), volumen(0:nxg-1,0:nyg-1,nzg))
        allocate(griduncn(0:nxg-1,0:nyg-1,nzg,maxspec,maxpointspec_act,nclassunc,maxageclass))
        allocate(wetgriduncn(0:nxg-1,0:nyg-1,maxspec,maxpointspec_act,nclassunc,maxageclass))
        allocate(drygriduncn(0:nxg-1,0:nyg-1,maxspec,maxpointspec_act,nclassunc,maxageclass))
        griduncn=0.; wetgriduncn=0.; drygriduncn=0.
      end if
    case ('outgeom')   ! dxout dyout outlon0 outlat0 outnum
      dxout=dbuf(1); dyout=dbuf(2); outlon0=dbuf(3); outlat0=dbuf(4); outnum=dbuf(5)
      dxoutn=dxout; dyoutn=dyout; outlon0n=outlon0; outlat0n=outlat0
    case ('outheight'); outheight(1:n)=dbuf(1:n)
    ! mixing-ratio output (iout = 2, 3): the met grid the air density comes from, concoutput.f90:176-205
    case ('iout');      iout=ibuf(1)
    case ('met')
      nx=ibuf(1); ny=ibuf(2); nz=ibuf(3); nxmin1=nx-1; nymin1=ny-1
    case ('metgeom');   dx=dbuf(1); dy=dbuf(2); xlon0=dbuf(3); ylat0=dbuf(4)
    case ('height');    height(1:n)=dbuf(1:n)
    case ('weightmolar'); weightmolar(1:n)=dbuf(1:n)
    case ('rho2')       ! rho(:,:,:,memind(2)), compact [nz][ny][nx]
      rho=0.
      do kz=1,nz
        do jy=0,ny-1
          do ix=0,nx-1
            rho(ix,jy,kz,2)=dbuf(1+ix+nx*(jy+ny*(kz-1)))
          end do
        end do
      end do
    case ('area')
      do jy=0,nyg-1
        do ix=0,nxg-1
          area(ix,jy)=dbuf(1+ix+nxg*jy)
        end do
      end do
    case ('volume')
      do kz=1,nzg
        do jy=0,nyg-1
          do ix=0,nxg-1
            volume(ix,jy,kz)=dbuf(1+ix+nxg*(jy+nyg*(kz-1)))
          end do
        end do
      end do
    case ('classes')   ! uncertainty classes the grids below carry: must be the nclassunc this binary was built with (par_mod)
      ncls=ibuf(1)
      if (ncls .ne. nclassunc) then
        write(*,*) 'ref_co_driver: the scenario has ', ncls, ' classes, this build nclassunc = ', nclassunc; stop 1
      end if
    case ('gridunc')   ! [classes][nspec][nzg][nyg][nxg]
      do kc=1,ncls
      do ks=1,nspec
        do kz=1,nzg
          do jy=0,nyg-1
            do ix=0,nxg-1
              gridunc(ix,jy,kz,ks,1,kc,1)=dbuf(1+ix+nxg*(jy+nyg*((kz-1)+nzg*((ks-1)+nspec*(kc-1)))))
            end do
          end do
        end do
      end do
      end do
    case ('wetgridunc')
      do kc=1,ncls
      do ks=1,nspec
        do jy=0,nyg-1
          do ix=0,nxg-1
            wetgridunc(ix,jy,ks,1,kc,1)=dbuf(1+ix+nxg*(jy+nyg*((ks-1)+nspec*(kc-1))))
          end do
        end do
      end do
      end do
    case ('drygridunc')
      do kc=1,ncls
      do ks=1,nspec
        do jy=0,nyg-1
          do ix=0,nxg-1
            drygridunc(ix,jy,ks,1,kc,1)=dbuf(1+ix+nxg*(jy+nyg*((ks-1)+nspec*(kc-1))))
          end do
        end do
      end do
      end do
    case default
      write(*,*) 'ref_co_driver: unknown record ', trim(name); stop 1
    end select
  end do
  close(uin)

  if (use_nest .eq. 1) then
    arean=area; volumen=volume
    griduncn=gridunc; wetgriduncn=wetgridunc; drygriduncn=drygridunc
    call concoutput_nest(itime_out, outnum)
  else
    call concoutput(itime_out, outnum, gtu, wtu, dtu)
  end if
end program coref
