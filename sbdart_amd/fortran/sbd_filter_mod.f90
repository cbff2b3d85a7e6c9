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

  ! Where the sun stands for an observer (zensun, spectra.f:4440-4556): zenith and azimuth angle in degrees and the
  ! Earth-Sun distance factor 1/r^2 for day of year iday, UTC hours `time`, latitude and longitude in degrees.
  ! Two points on the unit sphere -- the observer and the sub-solar point, each as (colatitude, longitude) -- and the
  ! spherical triangle they form with the pole; the sub-solar point's declination and the equation of time come from
  ! a table of five-day values.  (The sums keep the reference's operand order: the angles print to four digits, but
  ! they also enter the band model's air masses, which are compared bit for bit.)
  subroutine solar_position(iday, time, alat, alon, zenith, azimuth, solfac)
    integer, intent(in) :: iday
    real(kr), intent(in) :: time, alat, alon
    real(kr), intent(out) :: zenith, azimuth, solfac
    real(kr), parameter :: pi = 3.1415926536_kr, rad = pi/180.
    real(kr), parameter :: orbit_deg_per_day = 360./365.242, eccentricity = 0.01671, perihelion_day = 2.
    real(kr) :: day, minutes_fast, declination, colat_obs, colat_sun, dlon, cosz, east, north, distance
    day = mod(iday - 1, 365) + 1
    minutes_fast = five_day_table(tbl('sun.eqt'), day)
    declination = five_day_table(tbl('sun.dec'), day)
    colat_obs = (90. - alat)*rad
    colat_sun = (90. - declination)*rad
    dlon = (-15.*(time - 12. + minutes_fast/60.))*rad - alon*rad          ! sub-solar longitude minus the observer's
    cosz = cos(colat_obs)*cos(colat_sun) + sin(colat_obs)*sin(colat_sun)*cos(dlon)
    east = sin(colat_sun)*sin(dlon)
    north = sin(colat_obs)*cos(colat_sun) - cos(colat_obs)*sin(colat_sun)*cos(dlon)
    azimuth = atan2(east, north)/rad
    zenith = acos(cosz)/rad
    distance = 1. - eccentricity*cos(orbit_deg_per_day*(day - perihelion_day)*rad)
    solfac = 1./distance**2
  contains
    ! linear interpolation in a table whose entries stand for days 1, 6, 11, ... 366
    real(kr) function five_day_table(tab, d) result(v)
      real(kr), intent(in) :: tab(:), d
      integer :: hi
      real(kr) :: w
      hi = 2
      do while (1 + 5*(hi - 1) <= d .and. hi < 74)
        hi = hi + 1
      end do
      w = (d - (1 + 5*(hi - 2)))/real(5, kr)
      v = tab(hi - 1)*(1. - w) + w*tab(hi)
    end function
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
