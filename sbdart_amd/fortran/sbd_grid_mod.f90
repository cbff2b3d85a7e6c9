! Spectral grid and viewing geometry of a run, as objects.
!
! What they must reproduce (the interface of the reference, not its code layout):
!  * grid size: setfilt's rule (spectra.f:3370-3384) for the three meanings of WLINC
!    (> 1: constant wavenumber step in cm-1; < 0: constant step in ln(wavelength); else
!    constant wavelength step in um, 0 = "(wlsup-wlinf)/max(10, 1+int(range/0.005))");
!  * point il (0-based): central wavelength and the band edges handed to DISORT as WVNMLO/WVNMHI
!    (wllimits, drt.f:1657-1740): edges half a step either side, the first and last point
!    keep half a band, a lone point is +-0.0005 um wide;
!  * viewing angles (vuangles, drt.f:813-889): NPHI/NZEN > 0 expand two end values into an even
!    ramp (zeniths within 0.05 deg of the horizon are dropped), VZEN gives nadir angles, unset
!    arrays get the IOUT-dependent default grids.
! The arithmetic (operation order, REAL*4 literals) is the reference's, so that the band edges
! agree with the captured DISORT records to the last bit (sbd_grid_selftest).
module sbd_grid_mod
  implicit none
  private
  public :: kr, mxly, nstrms, unset, spectral_grid, new_grid, new_grid_from_bands, view_geometry, new_view

  integer, parameter :: kr = selected_real_kind(10)
  integer, parameter :: mxly = 65, nstrms = 40          ! params.f:9-11
  real(kr), parameter :: unset = -1._kr                 ! "zip": the namelist's not-set value (params.f:24)
  integer, parameter :: by_wavelength = 0, by_log = 1, by_wavenumber = 2, from_bands = 3

  type :: spectral_grid
    integer :: n = 1                 ! number of spectral points
    integer :: spacing = by_wavelength
    real(kr) :: lo = 0, hi = 0       ! first / last central wavelength (um)
    real(kr) :: step = 0             ! WLINC after the default rule
    ! from_bands (KDIST = -1): the spectral points are the bands of a k-distribution file, in file order
    real(kr), allocatable :: band_wl(:), band_lo(:), band_hi(:)
  contains
    procedure :: centre => grid_centre
    procedure :: band => grid_band
  end type

  type :: view_geometry
    integer :: nphi = 0, nzen = 0
    real(kr) :: phi(nstrms) = unset, uzen(nstrms) = unset
  end type

contains

  function new_grid(wlinf, wlsup, wlinc) result(g)
    real(kr), intent(in) :: wlinf, wlsup, wlinc
    type(spectral_grid) :: g
    real(kr) :: span
    g%lo = wlinf
    g%hi = wlsup
    g%step = wlinc
    span = wlsup - wlinf
    if (wlinf == wlsup) then                 ! a single spectral point, 0.001 um wide (spectra.f:3280-3298)
      g%spacing = by_wavelength
      g%n = 1
      g%step = real(.001, kr)
      return
    end if
    if (wlinc > 1._kr) then
      g%spacing = by_wavenumber
      g%n = int(((10000._kr/wlinf) - (10000._kr/wlsup))/wlinc + 1._kr)
    else if (wlinc < 0._kr) then
      g%spacing = by_log
      g%n = int(1 + log(wlsup/wlinf)/abs(wlinc))
    else
      g%spacing = by_wavelength
      if (wlinc == 0._kr) g%step = span/max(10, 1 + int(span/real(0.005, kr)))
      g%n = nint(span/g%step) + 1
    end if
    if (wlinf /= wlsup .and. g%n == 1) g%n = 2
  end function

  ! a grid that IS a list of bands (CKTAU records, readk: taugas.f:7695-7835): wavelength, WVNMLO, WVNMHI per point
  function new_grid_from_bands(wl, wvlo, wvhi) result(g)
    real(kr), intent(in) :: wl(:), wvlo(:), wvhi(:)
    type(spectral_grid) :: g
    g%spacing = from_bands
    g%n = size(wl)
    g%band_wl = wl; g%band_lo = wvlo; g%band_hi = wvhi
    if (g%n > 0) then
      g%lo = minval(wl); g%hi = maxval(wl)
    end if
  end function

  ! wavelength at (possibly fractional) grid coordinate x, 0 <= x <= n-1
  pure function grid_centre(g, x) result(w)
    class(spectral_grid), intent(in) :: g
    real(kr), intent(in) :: x
    real(kr) :: w, frac
    select case (g%spacing)
    case (by_wavenumber)
      frac = x/(g%n - 1)
      w = g%lo*g%hi/((1._kr - frac)*g%hi + frac*g%lo)
    case (by_log)
      frac = x/(g%n - 1)
      w = g%lo*(g%hi/g%lo)**frac
    case default
      w = g%lo + x*g%step
    end select
  end function

  ! spectral point il (0-based): central wavelength and DISORT's wavenumber interval
  subroutine grid_band(g, il, wl, wvnmlo, wvnmhi)
    class(spectral_grid), intent(in) :: g
    integer, intent(in) :: il
    real(kr), intent(out) :: wl, wvnmlo, wvnmhi
    real(kr) :: x, edge_lo, edge_hi
    real(kr), parameter :: half = 0.5_kr
    if (g%spacing == from_bands) then
      wl = g%band_wl(il + 1); wvnmlo = g%band_lo(il + 1); wvnmhi = g%band_hi(il + 1)
      return
    end if
    x = real(il, kr)
    wl = g%centre(x)
    if (g%spacing == by_wavelength) then
      edge_lo = wl - half*g%step
      edge_hi = wl + half*g%step
    else
      edge_lo = g%centre(x - half)
      edge_hi = g%centre(x + half)
    end if
    if (g%n > 1) then                      ! the two end points keep the inner half of their band
      if (il == 0) edge_lo = wl
      if (il == g%n - 1) edge_hi = wl
    end if
    if (edge_lo == wl .and. edge_hi == wl) then      ! a lone point
      edge_lo = wl - real(.0005, kr)
      edge_hi = wl + real(.0005, kr)
    end if
    wvnmlo = 10000._kr/edge_hi
    wvnmhi = 10000._kr/edge_lo
  end subroutine

  ! ---- viewing geometry ------------------------------------------------------------
  pure integer function last_set(a, notset) result(k)      ! index of the last entry that was given
    real(kr), intent(in) :: a(:), notset
    integer :: i
    k = 0
    do i = 1, size(a)
      if (a(i) /= notset) k = i
    end do
  end function

  pure function ramp(a, b, m) result(v)                    ! m evenly spaced values a..b
    real(kr), intent(in) :: a, b
    integer, intent(in) :: m
    real(kr) :: v(m)
    integer :: i
    do i = 1, m
      v(i) = a + (i - 1)*(b - a)/real(m - 1)
    end do
  end function

  ! phi/uzen/vzen/nphi/nzen as read from &INPUT
  function new_view(iout, nphi_in, phi_in, nzen_in, uzen_in, vzen_in) result(v)
    integer, intent(in) :: iout, nphi_in, nzen_in
    real(kr), intent(in) :: phi_in(nstrms), uzen_in(nstrms), vzen_in(nstrms)
    type(view_geometry) :: v
    real(kr) :: lo, hi, cand(2*nstrms)
    integer :: i, nv, m
    v%phi = phi_in
    v%uzen = uzen_in
    ! azimuths
    if (nphi_in > 0) then
      if (last_set(phi_in, unset) /= 2) write(*, '(a)') 'Error in MAIN -- '// &
           'must specify exactly 2 values of phi when nphi is set'
      if (nphi_in > nstrms) write(*, '(a)') 'Error in Main -- specified nphi larger than nstrms'
      v%nphi = nphi_in
      v%phi(1:v%nphi) = ramp(min(phi_in(1), phi_in(2)), max(phi_in(1), phi_in(2)), v%nphi)
    else
      v%nphi = last_set(phi_in, unset)
      if (v%nphi == 0) then
        v%nphi = 19
        v%phi(1:19) = ramp(0._kr, 180._kr, 19)
      end if
    end if
    ! nadir angles, if given, define the zeniths (vzen is "unset" at 90)
    nv = last_set(vzen_in, 90._kr)
    v%uzen(1:nv) = 180._kr - vzen_in(1:nv)
    ! zeniths
    if (nzen_in > 0) then
      if (last_set(v%uzen, unset) /= 2) write(*, '(a)') 'Error in MAIN -- '// &
           'must specify exactly 2 values of uzen when nzen is set'
      if (nzen_in > 2*nstrms) write(*, '(a)') 'Error in Main -- specified nzen larger than nstrms'
      m = min(nzen_in, 2*nstrms)
      cand(1:m) = ramp(min(v%uzen(1), v%uzen(2)), max(v%uzen(1), v%uzen(2)), m)
      v%nzen = 0
      do i = 1, m                                     ! the horizon itself is not a usable direction
        if (abs(cand(i) - 90._kr) > real(.05, kr) .and. v%nzen < nstrms) then
          v%nzen = v%nzen + 1
          v%uzen(v%nzen) = cand(i)
        end if
      end do
    else
      v%nzen = last_set(v%uzen, unset)
      if (v%nzen == 0) then
        select case (iout)
        case (5, 20)                                  ! looking down from the top
          v%nzen = 18; lo = 0; hi = 85
        case (6, 21)                                  ! looking up from the surface
          v%nzen = 18; lo = 95; hi = 180
        case default
          v%nzen = 36; lo = 0; hi = 180
        end select
        do i = 1, v%nzen
          v%uzen(i) = lo + (hi - lo)*(i - 1)/real(v%nzen - 1)
        end do
      end if
    end if
  end function

end module sbd_grid_mod
