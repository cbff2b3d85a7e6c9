! Gas absorption per wavelength: LOWTRAN7 20 cm-1 band model + continua + 3-term k-distribution with
! slant-path correction (reference: taugas.f:1802-1920 kdistr, 2236-2534 taugas and its spectral
! look-ups 2538-6821, 6940-7176, 7392-7510 gasset, 7650-7692 taucor).  Host side of SURVEY 8f row N1.
!
! Split the other way round from the reference: everything that depends on the wavelength only
! (continuum cross-sections, the band each molecule is in and its parameters) is evaluated ONCE per
! wavelength into a gas_spectrum; the two path evaluations (vertical, slant) and the k-distribution
! take it as input.  No saved state.  Literals: see sbd_tables_mod.
module sbd_gas_mod
  use sbd_grid_mod, only: kr
  use sbd_tables_mod
  implicit none
  private
  public :: gas_spectrum, spectrum_at, path_depths, gas_terms, nmol, mk, gas_tables_init

  integer, parameter :: nmol = 11          ! h2o co2 o3 n2o co ch4 o2 no so2 no2 nh3
  integer, parameter :: mk = 3             ! k-distribution terms
  character(len=3), parameter :: mol_name(nmol) = (/'h2o', 'co2', 'o3 ', 'n2o', 'co ', 'ch4', 'o2 ', 'no ', 'so2', 'no2', 'nh3'/)
  ! first absorber-amount slot of each molecule's bands minus one (slot - base = band number)
  integer, parameter :: slot_base(nmol) = (/16, 35, 30, 46, 43, 45, 49, 53, 55, 54, 51/)
  ! LOWTRAN7 band limits: molecule, first and last wavenumber (cm-1, 5 cm-1 grid), absorber-amount slot
  integer, parameter :: nrange = 64
  integer, parameter :: band_range(4, nrange) = reshape((/ &
    1, 0, 345, 17,   1, 350, 1000, 18,   1, 1005, 1640, 19,   1, 1645, 2530, 20,   1, 2535, 3420, 21, &
    1, 3425, 4310, 22,   1, 4315, 6150, 23,   1, 6155, 8000, 24,   1, 8005, 9615, 25,   1, 9620, 11540, 26, &
    1, 11545, 13070, 27,   1, 13075, 14860, 28,   1, 14865, 16045, 29,   1, 16340, 17860, 30, &
    3, 0, 200, 31,   3, 515, 1275, 32,   3, 1630, 2295, 33,   3, 2670, 2845, 34,   3, 2850, 3260, 35, &
    2, 425, 835, 36,   2, 840, 1440, 37,   2, 1805, 2855, 38,   2, 3070, 3755, 39,   2, 3760, 4065, 40, &
    2, 4530, 5380, 41,   2, 5905, 7025, 42,   2, 7395, 7785, 43,   2, 8030, 8335, 43,   2, 9340, 9670, 43, &
    5, 0, 175, 44,   5, 1940, 2285, 45,   5, 4040, 4370, 45,   6, 1065, 1775, 46,   6, 2345, 3230, 46, &
    6, 4110, 4690, 46,   6, 5865, 6135, 46,   4, 0, 120, 47,   4, 490, 775, 48,   4, 865, 995, 48, &
    4, 1065, 1385, 48,   4, 1545, 2040, 48,   4, 2090, 2655, 48,   4, 2705, 2865, 49,   4, 3245, 3925, 49, &
    4, 4260, 4470, 49,   4, 4540, 4785, 49,   4, 4910, 5165, 49,   7, 0, 265, 50,   7, 7650, 8080, 51, &
    7, 9235, 9490, 51,   7, 12850, 13220, 51,   7, 14300, 14600, 51,   7, 15695, 15955, 51, &
    7, 49600, 52710, 51,   11, 0, 385, 52,   11, 390, 2150, 53,   8, 1700, 2005, 54,   10, 580, 925, 55, &
    10, 1515, 1695, 55,   10, 2800, 2970, 55,   9, 0, 185, 56,   9, 400, 650, 57,   9, 950, 1460, 57, &
    9, 2415, 2580, 57 /), (/4, nrange/))

  ! the tables, resolved once per run by gas_tables_init (no name look-ups in the wavelength loop)
  type rtab
    real(kr), pointer :: p(:) => null()
  end type
  type itab
    integer, pointer :: p(:) => null()
  end type
  type(rtab), save :: t_self296, t_self260, t_foreign, t_n2, t_h1, t_h2, t_h3, t_o2s0, t_o2a, t_o2b, t_o4, t_o3uv, &
                      t_hh0, t_hh1, t_hh2, t_chap, t_schrun, t_cp(nmol), t_bs(nmol), t_ba(nmol), t_bb(nmol), t_bc(nmol)
  type(itab), save :: t_lo(nmol), t_hi(nmol)

  type gas_spectrum
    real(kr) :: v = 0                       ! wavenumber 1e4/wl
    ! continua (per unit of the absorber amounts of sbd_atmos_mod)
    real(kr) :: self296 = 0, self260 = 0, foreign = 0, radfn0 = 0, radfn1 = 0, far_wing = 0
    real(kr) :: n2 = 0, hno3 = 0, o2_herzberg = 0, o2_s = 0, o2_a = 0, o2_b = 0, o4 = 0
    real(kr) :: oz(3) = 0                   ! ozone: constant, linear and quadratic in (T - 273.15)
    ! band model
    integer :: slot(nmol) = -1              ! absorber-amount slot of the band each molecule is in, -1 = none
    real(kr) :: cp(nmol) = -20.             ! log10 absorption coefficient
    real(kr) :: bs(nmol) = 0, ba(nmol) = 0, bb(nmol) = 0, bc(nmol) = 0
  end type

contains

  subroutine gas_tables_init()
    integer :: m
    t_self296%p => tbl('h2o.self296'); t_self260%p => tbl('h2o.self260'); t_foreign%p => tbl('h2o.foreign')
    t_n2%p => tbl('n2.cont'); t_h1%p => tbl('hno3.h1'); t_h2%p => tbl('hno3.h2'); t_h3%p => tbl('hno3.h3')
    t_o2s0%p => tbl('o2.s0'); t_o2a%p => tbl('o2.a'); t_o2b%p => tbl('o2.b'); t_o4%p => tbl('o4.sig')
    t_o3uv%p => tbl('o3.uv'); t_hh0%p => tbl('o3.hh0'); t_hh1%p => tbl('o3.hh1'); t_hh2%p => tbl('o3.hh2')
    t_chap%p => tbl('o3.chappuis'); t_schrun%p => tbl('o2.schrun')
    do m = 1, nmol
      t_cp(m)%p => tbl('cp.'//trim(mol_name(m)))
      t_lo(m)%p => tbl_int('iwl.'//trim(mol_name(m))); t_hi(m)%p => tbl_int('iwh.'//trim(mol_name(m)))
      t_bs(m)%p => tbl('bs.'//trim(mol_name(m))); t_ba(m)%p => tbl('ba.'//trim(mol_name(m)))
      t_bb(m)%p => tbl('bb.'//trim(mol_name(m))); t_bc(m)%p => tbl('bc.'//trim(mol_name(m)))
    end do
  end subroutine

  ! value of a 10 cm-1 table that starts at v1: the entry at v, or the mean of the two around it when v
  ! is not a multiple of 10 (the reference's rule, taugas.f:3854-3871)
  real(kr) function ten_wavenumber_table(t, v1, v) result(c)
    real(kr), intent(in) :: t(:), v1, v
    integer :: i
    c = 0.
    i = (v - v1)/10. + 1.00001
    if (i >= size(t)) return
    c = t(i)
    if (mod(int(v), 10) > 0) c = (t(i) + t(i + 1))/2.
  end function

  ! band-model coefficient of one molecule at wavenumber v (5 cm-1 table over the molecule's spectral
  ! regions iwl..iwh, stored back to back); -20 outside every region (taugas.f:6416-6456)
  real(kr) function band_coefficient(m, v) result(c)
    integer, intent(in) :: m
    real(kr), intent(in) :: v
    integer, pointer :: lo(:), hi(:)
    real(kr), pointer :: cp(:)
    integer :: iv, r, before
    lo => t_lo(m)%p; hi => t_hi(m)%p; cp => t_cp(m)%p
    iv = v
    c = -20.0
    before = 0
    do r = 1, size(lo)
      if (lo(r) == -999) exit
      if (iv >= lo(r) .and. iv <= hi(r)) then
        c = cp(before + (iv - lo(r))/5 + 1)
        return
      end if
      before = before + (hi(r) - lo(r))/5 + 1
    end do
  end function

  function spectrum_at(wl, xo4) result(s)
    real(kr), intent(in) :: wl, xo4
    type(gas_spectrum) :: s
    real(kr), parameter :: bigexp = 87., fraco2 = .209, fracn2 = .781, effn2 = .2
    real(kr), pointer :: t(:), t1(:), t2(:)
    real(kr) :: v, xh2o, alpha2, xd, ya, yb, corr, rlosch, yratio, a, b, c, wnm, f, factor, vr, am, c0, xi
    integer :: iv5, i, m, n, k, inm
    v = 10000./wl
    iv5 = 5*(int(10000.0/wl)/5)
    s%v = v
    ! ---- water-vapour continuum: self at 296 K and 260 K, foreign; radiation field factors ----
    s%self296 = ten_wavenumber_table(t_self296%p, -20._kr, v)
    s%self260 = ten_wavenumber_table(t_self260%p, -20._kr, v)
    s%foreign = ten_wavenumber_table(t_foreign%p, -20._kr, v)
    if (s%self296 > 0.) then
      alpha2 = 200.**2
      xh2o = (1. - 0.2333*(alpha2/((v - 1050.)**2 + alpha2)))
      s%self296 = s%self296*xh2o
      s%self260 = s%self260*xh2o
    end if
    if ((v/0.6952)/260._kr <= bigexp) then
      xd = exp(-v/(296._kr*0.6952))
      s%radfn0 = v*(1. - xd)/(1. + xd)
      xd = exp(-v/(260._kr*0.6952))
      s%radfn1 = v*(1. - xd)/(1. + xd)
    else
      s%radfn0 = v
      s%radfn1 = v
    end if
    ya = exp(-log(1.025*3.159e-8) + (2.75e-4)*v)
    yb = exp(-log(8.97e-6) + (1.300e-3)*v)
    s%far_wing = 1./(ya + yb)
    ! ---- nitrogen continuum, 2080-2740 cm-1 ----
    if (v >= 2080. .and. v <= 2740.) then
      t => t_n2%p
      i = v
      s%n2 = t((i - 2080)/5 + 1)
    end if
    ! ---- nitric acid, three windows ----
    if (v >= 850.0 .and. v <= 920.0) then
      t => t_h1%p; i = (v - 845.)/5.; s%hno3 = t(i)
    else if (v >= 1275.0 .and. v <= 1350.0) then
      t => t_h2%p; i = (v - 1270.)/5.; s%hno3 = t(i)
    else if (v >= 1675.0 .and. v <= 1735.0) then
      t => t_h3%p; i = (v - 1670.)/5.; s%hno3 = t(i)
    end if
    ! ---- oxygen: Herzberg continuum (analytic), 1395-1760 cm-1 collision-induced band ----
    if (v > 36000.00) then
      corr = 0.
      if (v <= 40000.) corr = ((40000. - v)/4000.)*7.917e-27
      rlosch = 2.6868e24*1.0e-5
      yratio = v/48811.0
      s%o2_herzberg = (6.884e-24*(yratio)*exp(-69.738*(log(yratio))**2) - corr)*rlosch
    end if
    if (.not. (v < 1395 .or. v > 1760)) then
      i = (v - 1395.0_kr)/5.0_kr + 1.00001
      a = 0.; b = 0.; c = 0.
      t => t_o2s0%p
      if (i >= 1 .and. i <= size(t)) then
        c = t(i)
        t1 => t_o2a%p; t2 => t_o2b%p
        a = t1(i); b = t2(i)
      end if
      s%o2_a = a
      s%o2_b = a**2/2. + b
      s%o2_s = c/0.20946
    end if
    ! ---- O2-O2 / O2-N2 collision complexes, 1 nm table from 335 nm ----
    wnm = 1000.*wl
    inm = wnm
    f = wnm - inm
    inm = inm - 335 + 1
    if (inm >= 1 .and. inm <= 1015) then
      t => t_o4%p
      factor = fraco2**2
      if (wl > 1.2) factor = fraco2*(fraco2 + effn2*fracn2)
      s%o4 = xo4*factor*(t(inm)*(1. - f) + t(inm + 1)*f)
    end if
    ! ---- ozone: Hartley (UV), Hartley-Huggins with temperature terms, Chappuis ----
    if (v > 40800) then
      t => t_o3uv%p
      n = size(t)
      c = 0.
      i = (v - 40800._kr)/100._kr + 1.00001
      if (i >= 1 .and. i <= n) then
        vr = i*100._kr + 40800._kr
        if (vr <= (v + .1) .and. vr >= (v - .1)) then
          c = t(i)
        else
          if (i == n) i = n - 1
          am = (t(i + 1) - t(i))/100._kr
          c0 = t(i) - am*vr
          c = am*v + c0
        end if
      end if
      s%oz(1) = .269*c
    else if (v > 24370) then
      t => t_hh0%p
      i = (v - 27370._kr)/5._kr + 1.00001
      if (i >= 1 .and. i <= size(t)) then
        t1 => t_hh1%p; t2 => t_hh2%p
        s%oz(1) = .269*t(i)
        s%oz(2) = t(i)*t1(i)
        s%oz(3) = t(i)*t2(i)
      end if
    else if (v >= 13000. .and. v <= 24200) then
      t => t_chap%p
      xi = (v - 13000.0)/200.0 + 1.
      n = xi + 1.001
      s%oz(1) = t(n) + (xi - float(n))*(t(n) - t(n - 1))
    end if
    ! ---- band model: coefficient, band and band parameters of every molecule ----
    do m = 1, nmol
      s%cp(m) = band_coefficient(m, v)
    end do
    call assign_bands(band_range, nrange)
    if (iv5 >= 49600 .and. iv5 <= 52710) s%bs(7) = .4704          ! Schumann-Runge: its own band-model exponent
    if (v > 49600) then                                           ! ... and coefficients
      t => t_schrun%p
      s%cp(7) = -20.
      i = (v - 49600._kr)/5._kr + 1.0001
      if (i >= 1 .and. i <= size(t)) s%cp(7) = t(i)
    end if
  contains
    subroutine assign_bands(ranges, nr)
      integer, intent(in) :: nr, ranges(4, nr)
      integer :: band
      do k = 1, nr
        if (iv5 < ranges(2, k) .or. iv5 > ranges(3, k)) cycle
        m = ranges(1, k)
        s%slot(m) = ranges(4, k)
        band = ranges(4, k) - slot_base(m)
        s%bs(m) = t_bs(m)%p(band)
        s%ba(m) = t_ba(m)%p(band)
        s%bb(m) = t_bb(m)%p(band)
        s%bc(m) = t_bc(m)%p(band)
      end do
    end subroutine
  end function

  ! Continuum and band-model ("line") optical depth of every layer for a path whose zenith cosine at the
  ! ground is amu0 (spherical-shell air mass per layer); layer 1 is the top (taugas.f:2236-2534)
  subroutine path_depths(s, uu, amu0, z, nz, dtau_cont, dtau_line, column, carry)
    type(gas_spectrum), intent(in) :: s
    integer, intent(in) :: nz
    real(kr), intent(in) :: uu(mxq, nz), amu0, z(nz)
    real(kr), intent(out) :: dtau_cont(nz), dtau_line(nz)
    ! whole-path optical depth by absorber (IOUT 2): water lines, water continuum, CO2, O3, N2O, CO, CH4,
    ! O2 + N2 (lines and continua), trace gases (NO, SO2, NO2, NH3 lines)
    real(kr), intent(out), optional :: column(9)
    ! (the reference's report shows, for a molecule without a band at this wavelength, the value its work array
    !  still holds from an earlier wavelength -- and adds it to the total; `carry` reproduces that)
    real(kr), intent(inout), optional :: carry(nmol)
    real(kr) :: tau(nmol)
    real(kr), parameter :: awlmax = 20., wfac = 1.e-20
    real(kr) :: w(mxq), cum_c(nz), cum_l(nz), zi, zim, zbar, uniform, h2o, ozone, trace, awl
    integer :: i, im, k, ib
    cum_c = 0.; cum_l = 0.
    zim = z(nz)
    do i = nz, 1, -1
      im = nz - i + 1
      zi = z(i)
      zbar = 0.5*(zi + zim)
      zim = zi
      if (i == nz) then
        w = uu(:, i)/airmass_factor(zi)
      else
        w = w + (uu(:, i) - uu(:, i + 1))/airmass_factor(zbar)
      end if
      uniform = +s%o4*w(3) + s%n2*w(4) + s%o2_s*(w(63) + s%o2_a*(w(1) - 220*w(63)) + s%o2_b*w(2)) + s%o2_herzberg*w(58)
      h2o = s%self296*s%radfn0*(wfac*w(5)) + ((s%self260*s%radfn1) - (s%self296*s%radfn0))*(wfac*w(9)) &
            + (s%foreign + s%far_wing)*s%radfn0*(wfac*w(10))
      ozone = s%oz(1)*w(8) + s%oz(2)*w(59) + s%oz(3)*w(60)
      trace = s%hno3*w(11)
      cum_c(im) = uniform + h2o + ozone + trace
      tau = 0.
      if (present(carry)) tau = carry
      do k = 1, nmol
        ib = s%slot(k)
        if (ib > 0) then
          tau(k) = 0.
          if (s%cp(k) > -awlmax .and. w(ib) > 1.e-20) then
            awl = s%bs(k)*(s%cp(k) + log10(w(ib)))
            awl = min(awl, awlmax)
            tau(k) = 10.**awl
            cum_l(im) = cum_l(im) + tau(k)
          end if
        end if
      end do
      if (present(carry)) carry = tau
      if (present(column) .and. i == 1) column = (/tau(1), h2o, tau(2), tau(3) + ozone, tau(4), tau(5), tau(6), &
                                                    tau(7) + uniform, tau(8) + tau(9) + tau(10) + tau(11)/)
    end do
    dtau_cont(1) = cum_c(1)
    dtau_line(1) = cum_l(1)
    do i = 2, nz
      dtau_cont(i) = cum_c(i) - cum_c(i - 1)
      dtau_line(i) = cum_l(i) - cum_l(i - 1)
    end do
  contains
    real(kr) function airmass_factor(zz)
      real(kr), intent(in) :: zz
      airmass_factor = sqrt(1. - (1. - amu0**2)*(re_earth/(re_earth + zz))**2)
    end function
  end subroutine

  ! LOWTRAN7 three-term exponential-sum fit of the band transmission: per layer (1 = top) the optical
  ! depth increments of the three terms, their running sums, and the layer's weights (taugas.f:1802-1920)
  subroutine three_term_fit(s, uu, nz, dtk, tk, wtk)
    type(gas_spectrum), intent(in) :: s
    integer, intent(in) :: nz
    real(kr), intent(in) :: uu(mxq, nz)
    real(kr), intent(out) :: dtk(nz, mk), tk(nz, mk), wtk(nz, mk)
    real(kr), parameter :: fac(mk) = (/1.0, 0.09, 0.015/)
    real(kr) :: strength(mk, nmol), share(mk, nmol), cp1(nmol), duu, wpth, weighted, total
    integer :: k, m, n, lev, ib
    cp1 = 10.**s%cp
    strength = 0.; share = 0.
    do m = 1, nmol
      if (s%slot(m) > 0) then
        strength(:, m) = fac*s%bc(m)
        share(:, m) = (/s%ba(m), s%bb(m), 1. - s%ba(m) - s%bb(m)/)
      end if
    end do
    do k = 1, mk
      do n = 1, nz
        lev = nz - n + 1
        dtk(n, k) = 0.
        weighted = 0.
        do m = 1, nmol
          ib = s%slot(m)
          if (ib < 0) cycle
          if (lev == nz) then
            duu = uu(ib, lev)
          else
            duu = uu(ib, lev) - uu(ib, lev + 1)
          end if
          wpth = duu*strength(k, m)
          dtk(n, k) = dtk(n, k) + wpth*cp1(m)
          weighted = weighted + wpth*cp1(m)*share(k, m)
        end do
        wtk(n, k) = 1./3.
        if (dtk(n, k) /= 0) wtk(n, k) = weighted/dtk(n, k)
      end do
    end do
    do n = 1, nz
      total = wtk(n, 1) + wtk(n, 2) + wtk(n, 3)
      wtk(n, :) = wtk(n, :)/total
      if (n == 1) then
        tk(n, :) = dtk(n, :)
      else
        tk(n, :) = dtk(n, :) + tk(n - 1, :)
      end if
    end do
  end subroutine

  ! factor cf that makes the three-term transmission sum(g exp(-cf tau/amu)) equal exp(-utau): Newton
  ! iteration from cf = 1 (taugas.f:7650-7692)
  subroutine match_slant_transmission(gwk, tau, amu, utau, cf)
    real(kr), intent(in) :: gwk(mk), tau(mk), amu, utau
    real(kr), intent(out) :: cf
    real(kr) :: ff, f, fp
    integer :: it
    cf = 1.
    if (utau > 12.0) return
    do it = 1, 20
      ff = sum(gwk*exp(-cf*tau/amu))
      f = log(ff) + utau
      if (abs(f) < 0.000001) return
      fp = -sum(gwk*exp(-cf*tau/amu)*tau)/(ff*amu)
      cf = cf + (-f/fp)
    end do
    write(*, '(8es12.4)') gwk, tau, amu, utau
    stop 'TAUCOR: iteration did not converge'
  end subroutine

  ! The gas terms of one wavelength (gasset, taugas.f:7392-7510): number of k-terms nk (1 or 3), their
  ! weights, per layer the continuum depth and for each term the line depth -- columns 1..3 as fitted,
  ! columns 4..6 corrected to reproduce the slant-path band transmission at the solar zenith angle.
  subroutine gas_terms(kdist, s, uu, amu0, z, nz, nk, gwk, dtauk, dtau_cont)
    integer, intent(in) :: kdist, nz
    type(gas_spectrum), intent(in) :: s
    real(kr), intent(in) :: uu(mxq, nz), amu0, z(nz)
    integer, intent(out) :: nk
    real(kr), intent(out) :: gwk(mk), dtauk(nz, 2*mk), dtau_cont(nz)
    real(kr) :: dtcs(nz), dtls(nz), dtlv(nz), dtk(nz, mk), tk(nz, mk), wtk(nz, mk), wnorm, slant, cf
    real(kr) :: fitted(mk), corrected(mk)
    integer :: k, j
    call path_depths(s, uu, 1._kr, z, nz, dtau_cont, dtlv)
    if (amu0 > 0.) then
      call path_depths(s, uu, amu0, z, nz, dtcs, dtls)
    else
      dtls = dtlv
    end if
    gwk = 0.
    dtauk = 0.
    nk = 1
    gwk(1) = 1.
    if (.not. (kdist == 0 .or. sum(dtlv) < .01)) then
      call three_term_fit(s, uu, nz, dtk, tk, wtk)
      if (.not. max(tk(nz, 1), tk(nz, 2), tk(nz, 3)) < 0.01) then
        nk = mk
        do k = 1, mk
          gwk(k) = sum(dtlv*wtk(:, k))
        end do
        wnorm = sum(gwk)
        if (wnorm == 0) then
          gwk(1) = 1.
        else
          gwk = gwk/wnorm
        end if
      end if
    end if
    if (kdist == 0 .or. nk == 1) then
      dtauk(:, 1) = dtlv
      dtauk(:, 1 + mk) = amu0*dtls
    else
      dtauk(:, 1:mk) = dtk
      dtauk(:, mk + 1:2*mk) = dtk
      if (kdist >= 2 .and. amu0 > 0.) then
        slant = 0.
        corrected = 0.
        do j = 1, nz
          slant = slant + dtls(j)
          fitted = dtk(j, :)
          corrected = fitted + corrected
          call match_slant_transmission(gwk, corrected, amu0, slant, cf)
          dtauk(j, mk + 1:2*mk) = corrected*(cf - 1.0) + fitted
          corrected = cf*corrected
        end do
      end if
    end if
    if (amu0 <= 0.) dtauk(:, 1 + mk) = dtlv
  end subroutine

end module sbd_gas_mod
