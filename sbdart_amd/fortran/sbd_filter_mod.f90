! Sensor response ("filter") functions and the wavelength limits they imply (reference: setfilt
! spectra.f:3240-3387, filter 3390-3412, rdspec 4382-4416).  ISAT 0: flat between WLINF and WLSUP;
! -2 flat of width WLSUP about WLINF; -3 triangle, -4 Gaussian of equivalent width WLSUP about WLINF;
! -1 file filter.dat; 1..29 the built-in sensors (tables, data/TABLES.md).  Part of SURVEY 8f N1/N4.
module sbd_filter_mod
  use sbd_grid_mod, only: kr
  use sbd_tables_mod, only: tbl
  implicit none
  private
  public :: sensor_filter, new_filter, filter_value, read_spectrum_file, solar_position

  type sensor_filter
    integer :: n = 0                          ! 0: flat response
    real(kr) :: wlmin = 0, wlmax = 0
    real(kr), allocatable :: wl(:), resp(:)
  end type

contains

  ! Solar zenith and azimuth angle (degrees) and the Earth-Sun distance factor 1/r^2 for day of year iday, UTC
  ! hours `time`, latitude and longitude in degrees: equation of time and declination interpolated in a table of
  ! five-day values, spherical triangle pole-observer-subsolar point (zensun, spectra.f:4440-4556)
  subroutine solar_position(iday, time, alat, alon, zenith, azimuth, solfac)
    integer, intent(in) :: iday
    real(kr), intent(in) :: time, alat, alon
    real(kr), intent(out) :: zenith, azimuth, solfac
    real(kr), parameter :: pi = 3.1415926536_kr, degpday = 360./365.242, eccen = 0.01671, dayph = 2.
    real(kr), pointer :: eqt(:), dec(:)
    real(kr) :: dtor, dd, frac, eqtime, decang, sunlon, t0, t1, p0, p1, zz, xx, yy, rsun
    integer :: i, d0, d1
    eqt => tbl('sun.eqt'); dec => tbl('sun.dec')
    dtor = pi/180.
    dd = mod(iday - 1, 365) + 1
    i = 2                                            ! table days 1, 6, 11, ... 366: the first one beyond dd
    do while (1 + 5*(i - 1) <= dd .and. i < 74)
      i = i + 1
    end do
    d0 = 1 + 5*(i - 2); d1 = 1 + 5*(i - 1)
    frac = (dd - d0)/(d1 - d0)
    eqtime = eqt(i - 1)*(1. - frac) + frac*eqt(i)
    decang = dec(i - 1)*(1. - frac) + frac*dec(i)
    sunlon = -15.*(time - 12. + eqtime/60.)
    t0 = (90. - alat)*dtor
    t1 = (90. - decang)*dtor
    p0 = alon*dtor
    p1 = sunlon*dtor
    zz = cos(t0)*cos(t1) + sin(t0)*sin(t1)*cos(p1 - p0)
    xx = sin(t1)*sin(p1 - p0)
    yy = sin(t0)*cos(t1) - cos(t0)*sin(t1)*cos(p1 - p0)
    azimuth = atan2(xx, yy)/dtor
    zenith = acos(zz)/dtor
    rsun = 1. - eccen*cos(degpday*(dd - dayph)*dtor)
    solfac = 1./rsun**2
  end subroutine

  ! two-column text file "wavelength value", at most nmax lines, returned in ascending wavelength
  subroutine read_spectrum_file(file, nmax, wl, r)
    character(len=*), intent(in) :: file
    integer, intent(in) :: nmax
    real(kr), allocatable, intent(out) :: wl(:), r(:)
    real(kr) :: w(nmax), v(nmax)
    integer :: u, ios, i, n
    w = 0.
    open(newunit=u, file=file, status='old', form='formatted', iostat=ios)
    if (ios /= 0) then
      write(0, '(a)') 'sbdart_amd: cannot open '//file
      stop 2
    end if
    do i = 1, nmax
      read(u, *, iostat=ios) w(i), v(i)
      if (ios /= 0) exit
    end do
    close(u)
    n = count(w /= 0.)
    if (w(1) > w(n)) then
      wl = w(n:1:-1); r = v(n:1:-1)
    else
      wl = w(1:n); r = v(1:n)
    end if
  end subroutine

  function new_filter(isat, wlinf, wlsup) result(f)
    integer, intent(in) :: isat
    real(kr), intent(in) :: wlinf, wlsup
    type(sensor_filter) :: f
    real(kr), parameter :: pi = 3.1415926536_kr
    real(kr), pointer :: t(:)
    character(len=8) :: name
    real(kr) :: xlim, xx
    integer :: i
    if (wlsup == 0 .and. isat < -2) then
      write(*, *) 'Error -- WLSUP must be non-zero when ISAT=', isat
      write(*, *) '         WLSUP =', wlsup
      stop
    end if
    select case (isat)
    case (-4)
      xlim = 2*sqrt(pi)
      f%n = 1000
      f%wlmin = wlinf - 2*wlsup
      f%wlmax = wlinf + 2*wlsup
      allocate(f%wl(f%n), f%resp(f%n))
      do i = 1, f%n
        xx = -xlim + (i - 1)*(2*xlim)/(f%n - 1)
        f%resp(i) = exp(-xx**2)
        f%wl(i) = f%wlmin + (f%wlmax - f%wlmin)*real(i - 1)/(f%n - 1)
      end do
    case (-3)
      f%n = 3
      f%wlmin = wlinf - wlsup
      f%wlmax = wlinf + wlsup
      f%resp = (/0._kr, 1._kr, 0._kr/)
      f%wl = (/f%wlmin, wlinf, f%wlmax/)
    case (-2)
      f%wlmin = wlinf - .5*wlsup
      f%wlmax = wlinf + .5*wlsup
      if (wlsup == 0.) return
    case (-1)
      call read_spectrum_file('filter.dat', 5000, f%wl, f%resp)
      f%n = size(f%wl)
      f%wlmin = f%wl(1)
      f%wlmax = f%wl(f%n)
    case (0)
      f%wlmin = wlinf
      f%wlmax = wlsup
      if (wlinf == wlsup) return
    case default
      if (isat < 10) then
        write(name, '(a,i1)') 'filter', isat
      else
        write(name, '(a,i2)') 'filter', isat
      end if
      t => tbl(trim(name))
      f%n = size(t) - 2
      f%wlmin = t(1); f%wlmax = t(2)
      f%resp = t(3:)
      allocate(f%wl(f%n))
      do i = 1, f%n
        f%wl(i) = f%wlmin + (f%wlmax - f%wlmin)*real(i - 1)/(f%n - 1)
      end do
    end select
    if (f%wlmin < 0.199) then
      write(*, *) 'Error in SETFILT -- illegal wavelength limits '
      write(*, *) f%wlmin, f%wlmax
      stop
    end if
  end function

  real(kr) function filter_value(f, w) result(v)
    type(sensor_filter), intent(in) :: f
    real(kr), intent(in) :: w
    real(kr) :: wt
    integer :: lo, hi, mid
    if (f%n == 0) then
      v = 1.
      return
    end if
    if (w == f%wl(1)) then
      lo = 1
    else if (w == f%wl(f%n)) then
      lo = f%n - 1
    else
      lo = 1; hi = f%n
      do while (hi - lo > 1)
        mid = (hi + lo)/2
        if ((f%wl(f%n) > f%wl(1)) .eqv. (w > f%wl(mid))) then
          lo = mid
        else
          hi = mid
        end if
      end do
    end if
    wt = (w - f%wl(lo))/(f%wl(lo + 1) - f%wl(lo))
    wt = max(0._kr, min(1._kr, wt))
    v = f%resp(lo)*(1. - wt) + f%resp(lo + 1)*wt
  end function

end module sbd_filter_mod
