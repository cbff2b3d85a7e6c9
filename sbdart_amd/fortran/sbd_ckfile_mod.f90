! KDIST = -1: the gas optical depths of the run come from a correlated-k file pair instead of the band model.
!
!   CKATM (formatted):   nz, h2o density of the bottom level; z(1:nz); p(1:nz); t(1:nz) -- either order, stored
!                        bottom-up (gasinit, taugas.f:7297-7390)
!   CKTAU (unformatted, sequential): one record per sub-band, in order of DEcreasing wavenumber:
!                        iv, ib, nb, nk, vnu, vnu1, vnu2, etf, ewc (REAL*4), gw(1:nk), dtk(1:nz, 1:nk) (REAL*4)
!                        (readk, taugas.f:7695-7835).  ib counts the sub-bands of spectral point iv down from nb to 1;
!                        the output formats that print per point add the sub-bands up and print at ib = 1
!                        (drt.f:967-1044).
! The records inside the run's wavelength limits (setfilt's wl1, wl2; all of them when the two are equal) become
! the run's spectral points, in file order.
module sbd_ckfile_mod
  use sbd_grid_mod, only: kr, mxly
  implicit none
  private
  public :: ck_file, read_ck_files
  integer, parameter, public :: mxkd = 20               ! params.f:15

  type ck_file
    integer :: nz = 0
    real(kr) :: h2oden = 0
    real(kr), allocatable :: z(:), p(:), t(:)            ! bottom-up
    integer :: nrec = 0, npoint = 0, maxk = 1            ! sub-band records in range, spectral points (ib = 1), most k-terms
    integer, allocatable :: ib(:), nb(:), nk(:)
    real(kr), allocatable :: wl(:), wvlo(:), wvhi(:), etirr(:), ewcoef(:), gwk(:, :), dtauk(:, :, :)   ! (nz, mxkd, nrec)
  end type

contains

  subroutine read_ck_files(wllo, wlhi, ck)
    real(kr), intent(in) :: wllo, wlhi
    type(ck_file), intent(out) :: ck
    integer :: u, ios, iv, ib, nb, nk, pass, n
    real :: vnu0, vnu1, vnu2, etf, ewc, gw(mxkd)
    real, allocatable :: dtk(:, :)
    real(kr) :: vnulo, vnuhi
    open(newunit=u, file='CKATM', status='old', form='formatted', iostat=ios)
    if (ios /= 0) call die('kdist=-1: cannot open CKATM')
    read(u, *, iostat=ios) ck%nz, ck%h2oden
    if (ios /= 0 .or. ck%nz < 1) call die('kdist=-1: CKATM: bad first line (nz, h2o density)')
    if (ck%nz > mxly) then
      print *, 'gasinit --- nz gt mxly ', ck%nz, mxly
      stop
    end if
    allocate(ck%z(ck%nz), ck%p(ck%nz), ck%t(ck%nz))
    read(u, *, iostat=ios) ck%z
    if (ios == 0) read(u, *, iostat=ios) ck%p
    if (ios == 0) read(u, *, iostat=ios) ck%t
    if (ios /= 0) call die('kdist=-1: CKATM is truncated')
    close(u)
    if (ck%z(1) > ck%z(ck%nz)) then                     ! top-down in the file
      ck%z = ck%z(ck%nz:1:-1); ck%p = ck%p(ck%nz:1:-1); ck%t = ck%t(ck%nz:1:-1)
    end if
    if (wllo == wlhi) then
      vnulo = 0.
      vnuhi = huge(0.)
    else
      vnuhi = 10000./wllo
      vnulo = 10000./wlhi
    end if
    allocate(dtk(ck%nz, mxkd))
    do pass = 1, 2                                       ! count, then keep
      open(newunit=u, file='CKTAU', status='old', form='unformatted', iostat=ios)
      if (ios /= 0) call die('kdist=-1: cannot open CKTAU')
      n = 0
      do
        read(u, iostat=ios) iv, ib, nb, nk, vnu0, vnu1, vnu2, etf, ewc, gw(1:max(1, min(nk, mxkd))), dtk(1:ck%nz, 1:max(1, min(nk, mxkd)))
        if (ios /= 0) exit
        if (nk < 1 .or. nk > mxkd) call die('kdist=-1: CKTAU: number of k-terms out of range')
        if (vnu0 > vnuhi) cycle
        if (vnu0 < vnulo) exit
        n = n + 1
        if (pass == 2) then
          ck%ib(n) = ib; ck%nb(n) = nb; ck%nk(n) = nk
          ck%wvlo(n) = vnu1; ck%wvhi(n) = vnu2; ck%etirr(n) = etf; ck%ewcoef(n) = ewc
          ck%wl(n) = 10000./vnu0
          ck%gwk(1:nk, n) = gw(1:nk)
          ck%dtauk(:, 1:nk, n) = dtk(:, 1:nk)
          if (min(ck%wvlo(n), ck%wvhi(n)) <= 0.) then
            print *, 'readk --- wvnmlo,wvnmhi: ', ck%wvlo(n), ck%wvhi(n)
            stop
          end if
          if (minval(dtk(:, 1:nk)) < 0.) stop 'readk --- negative dtauk'
        end if
      end do
      close(u)
      if (pass == 1) then
        if (n == 0) then
          print *, 'Error --- gasinit'
          print *, 'no frequency samples within ', wllo, wlhi
          stop
        end if
        ck%nrec = n
        allocate(ck%ib(n), ck%nb(n), ck%nk(n), ck%wl(n), ck%wvlo(n), ck%wvhi(n), ck%etirr(n), ck%ewcoef(n), &
                 ck%gwk(mxkd, n), ck%dtauk(ck%nz, mxkd, n))
        ck%gwk = 0; ck%dtauk = 0
      end if
    end do
    ck%npoint = count(ck%ib == 1)
    ck%maxk = maxval(ck%nk)
  end subroutine

  subroutine die(msg)
    character(len=*), intent(in) :: msg
    write(0, '(a)') 'sbdart_amd: '//msg
    stop 1
  end subroutine

end module sbd_ckfile_mod
