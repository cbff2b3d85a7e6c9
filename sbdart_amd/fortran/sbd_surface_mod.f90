! Bidirectional surfaces of a run (ISALB 7, 8, 9): what the engine needs beside the work items.
!
! The reflectance models themselves live on the device (csrc/sbd_surface.hpp: BDREF's three models and SURFAC's
! azimuth quadrature, per work item and azimuth mode).  The host supplies
!   * the model and its run parameters from the namelist's SC array (suralb, spectra.f:61-177) -- for the ocean also
!     the two numbers seabdrf derives from the wind speed alone, foam cover and foam reflectance (spectra.f:441-451);
!   * for the ocean, per wavelength, the water's complex refractive index (indwat, spectra.f:594-1222: log-linear in
!     a table of pure water, plus a salinity term) and the sub-surface reflectance of case-I water (morcasiwat,
!     spectra.f:467-592: Morel's attenuation / backscattering coefficients on 400..700 nm, fixed-point iteration).
! Literals are written as the reference types them (default-real where it has default-real) so that the numbers
! handed to the engine are the reference's to the bit: tests/test_band_model.py compares them with the captured ones.
module sbd_surface_mod
  use iso_c_binding
  use sbd_grid_mod, only: kr
  use sbd_tables_mod, only: tbl
  use sbd_atmos_mod, only: bracket
  implicit none
  private
  public :: surface_model, new_surface_model, ocean_constants, flux_albedo

  type surface_model
    integer :: ibdrf = 0                 ! 0 Lambertian, 1 ocean, 2 Hapke, 3 Ross-thick / Li-sparse
    real(kr) :: par(8) = 0               ! sbd_run_cfg%bpar
    real(kr) :: pigment = 0              ! ocean: pigment concentration (mg/m3), SC(1)
    logical :: as_albedo = .false.       ! ISALB -7, -8, -9: a LAMBERTIAN surface with the model's flux albedo at the
                                         ! solar zenith angle (drt.f:478-484); the engine then sees ibdrf = 0
  end type

  interface
    function sbd_surface_flux_albedo(ibdrf, bpar, bitem, mu, albedo) bind(C, name='sbd_surface_flux_albedo') result(rc)
      import
      integer(c_int32_t), value :: ibdrf
      real(c_double), intent(in) :: bpar(8), bitem(4)
      real(c_double), value :: mu
      real(c_double), intent(out) :: albedo
      integer(c_int) :: rc
    end function
  end interface

contains

  function new_surface_model(isalb, sc) result(s)
    integer, intent(in) :: isalb
    real(kr), intent(in) :: sc(5)
    type(surface_model) :: s
    real(kr) :: cover
    s%as_albedo = isalb < 0
    select case (abs(isalb))
    case (7)                              ! SC = pigment, wind speed, salinity (suralb reads SC(3) as the salinity)
      s%ibdrf = 1
      s%pigment = sc(1)
      cover = real(2.951e-6, kr)*sc(2)**real(3.52, kr)         ! area covered by foam (Koepke 1984)
      s%par(1) = sc(2); s%par(2) = cover; s%par(3) = cover*real(0.22, kr); s%par(4) = sc(1); s%par(5) = sc(3)
    case (8)                              ! SC = single-scattering albedo, asymmetry, hot-spot amplitude and width
      s%ibdrf = 2
      s%par(1:4) = sc(1:4)
    case (9)                              ! SC = isotropic, volumetric, geometric coefficients, hot-spot magnitude, width
      s%ibdrf = 3
      s%par(1:5) = sc(1:5)
    end select
  end function

  ! DREF(mu): the model's flux albedo for incidence cosine mu -- the engine library's host-side integral over the
  ! model functions the device uses (sbd_surface_flux_albedo; disort.f:5178-5284)
  real(kr) function flux_albedo(s, bitem, mu) result(a)
    type(surface_model), intent(in) :: s
    real(kr), intent(in) :: bitem(4), mu
    real(c_double) :: par(8), bit(4), val
    integer(c_int) :: rc
    par = s%par; bit = bitem
    rc = sbd_surface_flux_albedo(int(s%ibdrf, c_int32_t), par, bit, real(mu, c_double), val)
    if (rc /= 0) then                                ! (disort.f:5262: |mu| > 1 -- cannot happen with a cosine)
      write(0, '(a)') 'sbdart_amd: DREF--input argument error(s)'
      stop 1
    end if
    a = val
  end function

  ! nr, ni of the water and the sub-surface reflectance rsw at wavelength wl (um); seabdrf hands the pigment
  ! concentration to BOTH look-ups -- to indwat in the place of the salinity (spectra.f:446-448)
  subroutine ocean_constants(s, wl, nr, ni, rsw)
    type(surface_model), intent(in) :: s
    real(kr), intent(in) :: wl
    real(kr), intent(out) :: nr, ni, rsw
    call water_index(wl, s%pigment, nr, ni)
    rsw = case1_water_reflectance(wl, s%pigment)
    if (s%pigment == 0.) rsw = 0.
  end subroutine

  subroutine water_index(wl, xsal, nr, ni)
    real(kr), intent(in) :: wl, xsal
    real(kr), intent(out) :: nr, ni
    real(kr), pointer :: wt(:), mr(:), mi(:)
    real(kr) :: f
    integer :: i
    real(kr), parameter :: nrc = 0.006, nic = 0.000
    wt => tbl('ocean.wl'); mr => tbl('ocean.mr'); mi => tbl('ocean.mi')
    i = bracket(wt, wl)
    f = (wl - wt(i))/(wt(i + 1) - wt(i))
    f = max(0.0_kr, min(1.0_kr, f))                  ! no extrapolation
    nr = mr(i)*(mr(i + 1)/mr(i))**f
    ni = mi(i)*(mi(i + 1)/mi(i))**f
    nr = nr + nrc*(xsal/34.3)
    ni = ni + nic*(xsal/34.3)
  end subroutine

  real(kr) function case1_water_reflectance(wl, c) result(rsw)
    real(kr), intent(in) :: wl, c
    real(kr), pointer :: tkw(:), txc(:), te(:), tbw(:)
    real(kr) :: kw, kd, xc, e, bw, bb, b, bbt, u1, r1, u2, err
    integer :: iwl
    if (wl < 0.400 .or. wl > 0.700) then
      rsw = 0.000
      return
    end if
    tkw => tbl('ocean.kw'); txc => tbl('ocean.xc'); te => tbl('ocean.e'); tbw => tbl('ocean.bw')
    iwl = 1 + nint((wl - 0.400)/0.005)
    kw = tkw(iwl); xc = txc(iwl); e = te(iwl); bw = tbw(iwl)
    if (abs(c) < 0.0001) then
      bb = 0.5*bw
      kd = kw
    else
      b = 0.30*c**0.62
      bbt = 0.002 + 0.02*(0.5 - 0.25*log10(c))*0.550/wl
      bb = 0.5*bw + bbt*b
      kd = kw + xc*c**e
    end if
    u1 = 0.75
    r1 = 0.33*bb/u1/kd
    do
      u2 = 0.90*(1. - r1)/(1. + 2.25*r1)
      rsw = 0.33*bb/u2/kd
      err = abs((rsw - r1)/rsw)
      if (err < 0.0001) exit
      r1 = rsw
    end do
  end function

end module sbd_surface_mod
