! Physical data tables of the band model (sbdart_amd/data/sbdart_tables.bin, written by
! tools/extract_tables.py from the compiled reference; provenance per table in data/TABLES.md) and the
! constants of the reference's params module (params.f:17-31).
!
! A word on literals, valid for every band-model module: the reference computes in real(kr) = fp64 but
! types most of its constants as default-real (fp32) literals, which Fortran widens.  The parity bar on
! the per-wavelength optical depths is 1e-12, so the same constants are typed the same way here: a bare
! literal in these modules is deliberately fp32.
module sbd_tables_mod
  use sbd_grid_mod, only: kr
  implicit none
  private
  public :: tables_load, tbl, tbl_int, tables_loaded, tables_image
  public :: pzero, tzero, re_earth, pmo, grav, alosch, mxq

  integer, parameter :: mxq = 63                       ! absorber-amount slots (params.f:14)
  real(kr), parameter :: pzero = 1013.25, tzero = 273.15, re_earth = 6371.2, pmo = 2.6568e-23, &
                         grav = 9.80665, alosch = 2.6868e19

  type entry_t
    character(len=24) :: name = ''
    real(kr), allocatable :: r(:)
    integer, allocatable :: i(:)
  end type
  type(entry_t), allocatable, target, save :: entries(:)
  logical, save :: tables_loaded = .false.
  ! the file as it is, for the engine's gas kernel (include/sbdart_amd.h, sbd_gas_model%tables)
  integer(kind=1), allocatable, target, save :: tables_image(:)

contains

  ! Looks for the table file in: $SBD_TABLES, <directory of the executable>/../data/, ./
  subroutine tables_load(ok, tried)
    logical, intent(out) :: ok
    character(len=*), intent(out) :: tried
    character(len=1024) :: path, exe
    character(len=8) :: magic
    integer :: plen, pstat, u, ios, ntab, k, kind, n, cut, ic
    integer(kind=4) :: pad
    ok = tables_loaded
    tried = ''
    if (ok) return
    do ic = 1, 3
      path = ''
      select case (ic)
      case (1)
        call get_environment_variable('SBD_TABLES', path, plen, pstat)
        if (pstat /= 0 .or. plen <= 0) cycle
      case (2)
        call get_command_argument(0, exe)
        cut = index(exe, '/', back=.true.)
        if (cut == 0) then
          path = '../data/sbdart_tables.bin'
        else
          path = exe(1:cut)//'../data/sbdart_tables.bin'
        end if
      case (3)
        path = 'sbdart_tables.bin'
      end select
      tried = trim(tried)//' '//trim(path)
      open(newunit=u, file=trim(path), access='stream', form='unformatted', status='old', iostat=ios)
      if (ios /= 0) cycle
      read(u, iostat=ios) magic, ntab
      if (ios /= 0 .or. magic(1:7) /= 'SBDTBL1') then
        close(u)
        cycle
      end if
      inquire(unit=u, size=n)
      if (n > 0) then
        allocate(tables_image(n))
        read(u, pos=1) tables_image
        read(u, pos=13)                                    ! (back behind the header)
      end if
      allocate(entries(ntab))
      do k = 1, ntab
        read(u) entries(k)%name, kind, n
        if (kind == 1) then
          allocate(entries(k)%r(n))
          read(u) entries(k)%r
        else
          allocate(entries(k)%i(n))
          read(u) entries(k)%i
          if (mod(n, 2) == 1) read(u) pad
        end if
      end do
      close(u)
      tables_loaded = .true.
      ok = .true.
      return
    end do
  end subroutine

  function find(name) result(k)
    character(len=*), intent(in) :: name
    integer :: k
    do k = 1, size(entries)
      if (trim(entries(k)%name) == name) return
    end do
    write(0, '(a)') 'sbdart_amd: table '//name//' is not in sbdart_tables.bin'
    stop 3
  end function

  function tbl(name) result(p)
    character(len=*), intent(in) :: name
    real(kr), pointer :: p(:)
    p => entries(find(name))%r
  end function

  function tbl_int(name) result(p)
    character(len=*), intent(in) :: name
    integer, pointer :: p(:)
    p => entries(find(name))%i
  end function

end module sbd_tables_mod
