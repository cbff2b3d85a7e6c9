! The atmosphere of a run: model profiles, the user's rescaling of water vapour / ozone / pressure,
! trace-gas mixing ratios, and the absorber amounts the LOWTRAN7 band model integrates over
! (reference: atms.f:223-420, 422-450; taugas.f:1924-2233, 7178-7295; drt.f:1181-1266).
! Host side of SURVEY 8f row N1; once per run.  Literals: see sbd_tables_mod.
module sbd_atmos_mod
  use sbd_grid_mod, only: kr, unset
  use sbd_tables_mod
  implicit none
  private
  public :: atmosphere, model_atmosphere, rescale_profiles, pressure_at, trace_gases, set_trace_gases, &
            absorber_columns, bracket, nearest_index, relative_humidity, regrid, user_atmosphere, mix_in, saturate_clouds

  type atmosphere                     ! levels bottom-up: index 1 is the surface
    integer :: nz = 0
    real(kr), allocatable :: z(:), p(:), t(:), wh(:), wo(:)    ! km, mb, K, g/m3 water vapour, g/m3 ozone
  end type

  integer, parameter :: mxly_levels = 65            ! most levels of a run (params.f:9)
  integer, parameter :: ngas = 11     ! n2 o2 co2 ch4 n2o co no2 so2 nh3 no hno3
  character(len=4), parameter :: gas_name(ngas) = &
    (/'n2  ', 'o2  ', 'co2 ', 'ch4 ', 'n2o ', 'co  ', 'no2 ', 'so2 ', 'nh3 ', 'no  ', 'hno3'/)
  type trace_gases                    ! surface-value scale factors of the standard mixing-ratio profiles
    real(kr) :: scale(ngas) = 1
    logical :: scaled = .false.
    real(kr) :: xo4 = 1               ! multiplier of the O4 collision continuum
  end type

contains

  ! index j (1..n-1) with xx(j) <= x < xx(j+1) for ascending xx (descending: mirrored); the ends clamp
  integer function bracket(xx, x) result(j)
    real(kr), intent(in) :: xx(:), x
    integer :: lo, hi, mid, n
    logical :: up
    n = size(xx)
    if (x == xx(1)) then
      j = 1
    else if (x == xx(n)) then
      j = n - 1
    else
      up = xx(n) > xx(1)
      lo = 1; hi = n
      do while (hi - lo > 1)
        mid = (hi + lo)/2
        if (up .eqv. (x > xx(mid))) then
          lo = mid
        else
          hi = mid
        end if
      end do
      j = lo
    end if
  end function

  integer function nearest_index(xx, x) result(j)
    real(kr), intent(in) :: xx(:), x
    j = bracket(xx, x)
    if (abs(x - xx(j + 1)) < abs(x - xx(j))) j = j + 1
  end function

  function model_atmosphere(idatm) result(a)
    integer, intent(in) :: idatm
    type(atmosphere) :: a
    real(kr), pointer :: t(:)
    character(len=4) :: name
    integer :: n
    write(name, '(a,i1)') 'atm', abs(idatm)
    t => tbl(name)
    n = size(t)/5
    a%nz = n
    allocate(a%z(n), a%p(n), a%t(n), a%wh(n), a%wo(n))
    a%z = t(1:n); a%p = t(n + 1:2*n); a%t = t(2*n + 1:3*n); a%wh = t(3*n + 1:4*n); a%wo = t(4*n + 1:5*n)
  end function

  ! IDATM = 0: the profile of atms.dat -- number of levels, then "z p t wh wo" per level, top level first
  ! (either order is accepted and turned bottom-up) (useratm, atms.f:452-503)
  function user_atmosphere() result(a)
    type(atmosphere) :: a
    integer :: u, ios, n, i
    open(newunit=u, file='atms.dat', status='old', form='formatted', iostat=ios)
    if (ios /= 0) then
      write(0, '(a)') 'sbdart_amd: cannot open atms.dat'
      stop 2
    end if
    read(u, *) n
    if (n > mxly_levels) then
      write(*, '(a,i3,a,i3)') 'error in USERATM: ', n, ' layers specified in ATMS.DAT, but current limit is ', mxly_levels
      stop
    end if
    a%nz = n
    allocate(a%z(n), a%p(n), a%t(n), a%wh(n), a%wo(n))
    do i = n, 1, -1
      read(u, *) a%z(i), a%p(i), a%t(i), a%wh(i), a%wo(i)
    end do
    close(u)
    if (a%z(1) > a%z(n)) then
      a%z = a%z(n:1:-1); a%p = a%p(n:1:-1); a%t = a%t(n:1:-1); a%wh = a%wh(n:1:-1); a%wo = a%wo(n:1:-1)
    end if
  end function

  ! AMIX: a weighted mean of the model atmosphere a (weight 1 - amix) and atms.dat (weight amix), level by
  ! level on the same altitudes (atms.f:436-448)
  subroutine mix_in(a, amix)
    type(atmosphere), intent(inout) :: a
    real(kr), intent(in) :: amix
    type(atmosphere) :: b
    b = user_atmosphere()
    if (b%nz /= a%nz) stop 'atms -- vertical grids do not match'
    if (any(abs(b%z - a%z) > 0.01)) stop 'atms -- vertical grids do not match'
    a%p = a%p*(1. - amix) + b%p*amix
    a%t = a%t*(1. - amix) + b%t*amix
    a%wh = a%wh*(1. - amix) + b%wh*amix
    a%wo = a%wo*(1. - amix) + b%wo*amix
  end subroutine

  ! NGRID /= 0: |ngrid| levels between the surface and the model top, spacing zgrid1 at the bottom,
  ! zgrid2 at the top, stretched by a power law in between; pressure and (positive) densities
  ! interpolated logarithmically, temperature linearly (zgrid, atms.f:505-575)
  subroutine regrid(a, zgrid1, zgrid2, ngrid)
    type(atmosphere), intent(inout) :: a
    real(kr), intent(in) :: zgrid1, zgrid2
    integer, intent(in) :: ngrid
    real(kr), parameter :: tol = 0.99
    type(atmosphere) :: b
    real(kr) :: span, beta, toprat, x, ztop, fz
    integer :: ng, i, j, jj, nz
    nz = a%nz
    ng = min(mxly_levels, abs(ngrid))
    span = zgrid1*float(ng - 1)
    ztop = a%z(nz)
    if (span >= ztop) then
      span = ztop
      beta = 0.
    else
      span = min(span, tol*(ztop - zgrid2))
      toprat = float(ng - 2)/(ng - 1)
      beta = log(((ztop - zgrid2)/(toprat*span) - 1.)/(ztop/span - 1.))/log(toprat)
    end if
    b%nz = ng
    allocate(b%z(ng), b%p(ng), b%t(ng), b%wh(ng), b%wo(ng))
    j = 2
    do i = 1, ng
      x = float(i - 1)/float(ng - 1)
      b%z(i) = span*x*(1. + (ztop/span - 1.)*x**beta)
      jj = j
      do while (jj <= nz)                    ! first old level at or above the new one, never going back down
        if (b%z(i) <= a%z(jj)) exit
        jj = jj + 1
      end do
      j = min(jj, nz)
      fz = (b%z(i) - a%z(j - 1))/(a%z(j) - a%z(j - 1))
      fz = min(max(fz, 0._kr), 1._kr)
      b%p(i) = a%p(j - 1)*(a%p(j)/a%p(j - 1))**fz
      b%t(i) = a%t(j - 1)*(1. - fz) + a%t(j)*fz
      b%wh(i) = density_between(a%wh(j - 1), a%wh(j))
      b%wo(i) = density_between(a%wo(j - 1), a%wo(j))
    end do
    a = b
  contains
    real(kr) function density_between(d1, d2) result(d)
      real(kr), intent(in) :: d1, d2
      if (min(d2, d1) > 0.) then
        d = d1*(d2/d1)**fz
      else
        d = d1*(1. - fz) + d2*fz
      end if
    end function
  end subroutine

  ! pressure at altitude zq, log-interpolated between the levels around it (drt.f:305-309)
  real(kr) function pressure_at(a, zq) result(pq)
    type(atmosphere), intent(in) :: a
    real(kr), intent(in) :: zq
    integer :: j
    real(kr) :: f
    j = bracket(a%z, zq)
    f = (zq - a%z(j))/(a%z(j + 1) - a%z(j))
    pq = a%p(j)*(a%p(j + 1)/a%p(j))**f
  end function

  ! column between two levels of a density that varies exponentially with height (log-mean), falling
  ! back to the trapezoid when the two densities are within 0.1 % or one is zero
  pure real(kr) function slab_column(dz, d1, d2) result(du)
    real(kr), intent(in) :: dz, d1, d2
    if (abs(d1 - d2) <= .001*d1 .or. min(d1, d2) == 0.) then
      du = .5*dz*(d1 + d2)
    else
      du = dz*(d1 - d2)/log(d1/d2)
    end if
  end function

  ! UW / SCLH2O, UO3 / O3TRP / ZTRP, PBAR (atms.f:223-420): negative = keep the model's own
  subroutine rescale_profiles(a, sclh2o, uw, uo3, o3trp, ztrp, pbar)
    type(atmosphere), intent(inout) :: a
    real(kr), intent(in) :: sclh2o, uw, uo3, o3trp, ztrp, pbar
    real(kr) :: total, ofac, above, below, edge, fac_above, fac_below, dz
    integer :: i, nz
    nz = a%nz
    if (uw >= 0.) then
      if (sclh2o > 0.) then
        a%wh = (uw/sclh2o)*exp(-a%z/sclh2o)
      else
        total = 0.
        do i = nz - 1, 1, -1
          total = total + slab_column(a%z(i + 1) - a%z(i), a%wh(i), a%wh(i + 1))
        end do
        total = 0.1*total                               ! g km / m3 -> g / cm2
        a%wh = (uw/total)*a%wh
      end if
    end if
    if (uo3 >= 0. .or. o3trp >= 0.) then
      ofac = .5/(30*pmo*alosch)
      above = 0.; below = 0.; edge = 0.
      do i = nz - 1, 1, -1
        dz = a%z(i + 1) - a%z(i)
        if (a%z(i) >= ztrp) then
          above = above + ofac*dz*(a%wo(i) + a%wo(i + 1))
        else if (edge == 0) then                        ! the slab that straddles ZTRP: its upper level scales with
          edge = ofac*dz*a%wo(i + 1)                    ! the column above, its lower level with the column below
          below = ofac*dz*a%wo(i)
        else
          below = below + ofac*dz*(a%wo(i) + a%wo(i + 1))
        end if
      end do
      fac_above = 1.; fac_below = 1.
      if (uo3 >= 0.) then
        if (above == 0) then
          print *, 'Error in modatm -- original ozone column density above ZTRP = 0  -- can not modify'
          stop
        end if
        fac_above = uo3/above
      end if
      if (o3trp >= 0) then
        if (below == 0) then
          print *, 'Error in modatm -- original ozone column density below ZTRP = 0  -- can not modify'
          stop
        end if
        fac_below = max(o3trp - edge*fac_above, 0._kr)/below
      end if
      where (a%z < ztrp)
        a%wo = fac_below*a%wo
      elsewhere
        a%wo = fac_above*a%wo
      end where
    end if
    if (pbar >= 0.) a%p = (pbar/a%p(1))*a%p
  end subroutine

  ! XN2 ... XHNO3 (surface volume mixing ratios, ppm) and XO4 (taugas.f:7256-7295)
  subroutine set_trace_gases(g, x, xo4)
    type(trace_gases), intent(out) :: g
    real(kr), intent(in) :: x(ngas), xo4
    real(kr), pointer :: prof(:)
    integer :: k
    do k = 1, ngas
      if (x(k) >= 0.) then
        prof => tbl('mix.'//trim(gas_name(k)))
        g%scale(k) = x(k)/prof(1)
      end if
    end do
    g%xo4 = xo4
    g%scaled = maxval(x) > -0.99
  end subroutine

  ! volume mixing ratios (ppm) of the eleven uniformly-mixed / trace gases at altitude zq (taugas.f:7178-7254)
  subroutine mixing_ratios(g, zq, vf)
    type(trace_gases), intent(in) :: g
    real(kr), intent(in) :: zq
    real(kr), intent(out) :: vf(ngas)
    real(kr), pointer :: alt(:), prof(:)
    real(kr) :: zc, f
    integer :: k, m
    alt => tbl('mix.alt')
    zc = max(0._kr, min(zq, 100._kr))
    k = 1                                    ! the table interval that holds zc (the last one includes its top)
    do m = 2, size(alt) - 1
      if (alt(m) <= zc) k = m
    end do
    f = (zc - alt(k))/(alt(k + 1) - alt(k))
    do m = 1, ngas
      prof => tbl('mix.'//trim(gas_name(m)))
      vf(m) = prof(k)*(1. - f) + prof(k + 1)*f
    end do
    if (g%scaled) vf = vf*g%scale
  end subroutine

  ! Absorber amounts from each level to space, uu(slot, level) (absint, taugas.f:1924-2233).  Slots:
  ! 1-2 O2 (temperature weighted), 3 O4, 4 N2 continuum, 5/9/10 H2O self (296 K, T-dependent) and foreign
  ! continua, 6 molecular (Rayleigh) column, 8 O3, 11 HNO3, 17-57 the band-model absorber bands with their
  ! pressure and temperature scaling, 58 O2 Herzberg, 59-60 O3 temperature terms, 63 O2.
  subroutine absorber_columns(a, g, uu)
    type(atmosphere), intent(in) :: a
    type(trace_gases), intent(in) :: g
    real(kr), intent(out) :: uu(mxq, a%nz)
    real(kr), parameter :: xlosch = alosch*1.e5, conjoe = 0.1/alosch, con = 3.3429e21, rhzero = tzero/296.0
    ! band slots 17..57: which gas (0 = water vapour, -1 = ozone, else index into the trace gases) and the
    ! exponents of p/p0 and T0/T of the LOWTRAN7 scaled absorber amount
    integer, parameter :: slot0 = 17, nslot = 41
    integer, parameter :: who(nslot) = (/ (0, integer :: i_ = 1, 14), (-1, integer :: i_ = 1, 5), (3, integer :: i_ = 1, 8), &
         6, 6, 4, 5, 5, 5, 2, 2, 9, 9, 10, 7, 8, 8 /)
    real, parameter :: pexp(nslot) = (/ 0.9810, 1.1406, 0.9834, 1.0443, 0.9681, 0.9555, 0.9362, 0.9233, 0.8658, 0.8874, &
         0.7982, 0.8088, 0.6642, 0.6656, 0.4200, 0.4221, 0.3739, 0.1770, 0.3921, 0.6705, 0.7038, 0.7258, 0.6982, 0.8867, &
         0.7883, 0.6899, 0.6035, 0.7589, 0.9267, 0.7139, 0.3783, 0.7203, 0.7764, 1.1879, 0.9353, 0.8023, 0.6968, 0.5265, &
         0.3956, 0.2943, 0.2135 /)
    real, parameter :: texp(nslot) = (/ 0.3324, -2.6343, -2.5294, -2.4359, -1.9537, -1.5378, -1.6338, -0.9398, -0.1034, &
         -0.2576, 0.0588, 0.2816, 0.2764, 0.5061, 1.3909, 0.7678, 0.1225, 0.9827, 0.1942, -2.2560, -5.0768, -1.6740, &
         -1.8107, -0.5327, -1.3244, -0.8152, 0.6026, 0.6911, 0.1716, -0.4185, 0.9399, -0.1836, 1.1931, 2.9738, 0.1936, &
         -0.9111, 0.3377, -0.4702, -0.0545, 1.2316, 0.0733 /)
    real(kr) :: dd(mxq, a%nz), vf(ngas), tt, pp, pss, tss, f1, f2, wair, rhoh2o, wo2d, vfo3, amount, cw
    real(kr) :: ztop, ptop, ttop, dz, tbar, dp, drho, den1, den2, above, dmin, dave, du, du_self, du_o3, tfac, scfac
    integer :: i, k, nz
    logical :: linear
    nz = a%nz
    uu = 0.
    if (maxval(a%p) == 0.) return
    dd = 0.
    do i = 1, nz
      call mixing_ratios(g, a%z(i), vf)
      tt = a%t(i); pp = a%p(i)
      pss = pp/pzero
      tss = tzero/tt
      f1 = (pp/pzero)/(tt/tzero)
      f2 = (pp/pzero)*sqrt(tzero/tt)
      wair = alosch*f1
      rhoh2o = con*a%wh(i)/xlosch
      wo2d = conjoe*wair*vf(2)*pss
      vfo3 = a%wo(i)/(3*pmo*wair)
      dd(1, i) = wo2d*tt
      dd(2, i) = wo2d*(tt - 220.)**2
      dd(3, i) = f1**2
      dd(4, i) = 1.e-6*vf(1)*f1*f2
      dd(5, i) = xlosch*rhoh2o**2/rhzero
      dd(6, i) = f1
      dd(8, i) = conjoe*wair*vfo3
      dd(10, i) = xlosch*rhoh2o*(f1 - rhoh2o)/rhzero
      dd(11, i) = f1*vf(11)*(1.e-6*1.e5)
      dd(63, i) = wo2d
      cw = conjoe*wair
      do k = 1, nslot
        select case (who(k))
        case (0)
          amount = a%wh(i)*.1
        case (-1)
          amount = cw*vfo3
        case default
          amount = cw*vf(who(k))
        end select
        dd(slot0 + k - 1, i) = amount*pss**real(pexp(k), kr)*tss**real(texp(k), kr)
      end do
      dd(58, i) = (1. + .83*f1)*(cw*vf(2))
    end do
    scfac = exp(-1.)
    do i = nz, 1, -1
      if (i == nz) then              ! the column above the top level: one more slab, densities falling by 1/e
        ztop = 2*a%z(i) - a%z(i - 1)
        ptop = a%p(i)**2/a%p(i - 1)
        ttop = a%t(i)
      else
        ztop = a%z(i + 1); ptop = a%p(i + 1); ttop = a%t(i + 1)
      end if
      dz = ztop - a%z(i)
      if (a%p(i) == ptop) then
        tbar = .5*(ttop + a%t(i))
      else                           ! density-weighted mean temperature of the slab
        dp = (a%p(i) - ptop)/log(a%p(i)/ptop)
        drho = (a%p(i)/a%t(i) - ptop/ttop)/log(a%p(i)*ttop/(ptop*a%t(i)))
        tbar = dp/drho
      end if
      du_self = 0; du_o3 = 0
      do k = 1, mxq
        den1 = dd(k, i)
        if (i == nz) then
          above = 0.
          den2 = dd(k, i)*scfac
        else
          above = uu(k, i + 1)
          den2 = dd(k, i + 1)
        end if
        linear = k == 8 .or. (k >= 31 .and. k <= 35) .or. k == 59 .or. k == 60      ! ozone: trapezoid
        dmin = min(den1, den2)
        dave = .5*(den1 + den2)
        if (dmin > 0. .and. dmin < 0.999*dave .and. .not. linear) then
          du = dz*(den1 - den2)/log(den1/den2)
        else
          du = dz*dave
        end if
        if (k == 5) du_self = du
        if (k == 8) du_o3 = du
        select case (k)
        case (9)                     ! part of the self continuum taken at 260 K
          tfac = (296. - tbar)/(296. - 260.)
          tfac = max(0._kr, min(1._kr, tfac))
          uu(9, i) = above + du_self*tfac
        case (59)
          uu(59, i) = above + .269*du_o3*(tbar - 273.15)
        case (60)
          uu(60, i) = above + .269*du_o3*(tbar - 273.15)**2
        case default
          uu(k, i) = above + du
        end select
      end do
    end do
  end subroutine

  ! RHCLD >= 0: water vapour inside the cloud layers (and one level above each cloud top) set to relative
  ! humidity rhcld.  KRHCLR = 1 leaves the clear levels alone (satcloud, atms.f:11-67); otherwise the clear
  ! levels are rescaled so that the column keeps its water vapour, or, if the clouds alone hold more than
  ! that, the cloud levels are (saturate, atms.f:70-221).  cloud_layer: layer numbers from the cloud slots
  ! (1 = top, negative = upper end of an extended cloud, 0 = unused).
  subroutine saturate_clouds(a, cloud_layer, rhcld, keep_clear)
    type(atmosphere), intent(inout) :: a
    integer, intent(in) :: cloud_layer(:)
    real(kr), intent(in) :: rhcld
    logical, intent(in) :: keep_clear
    real(kr) :: column, clear, cloudy, cldfac, clrfac, after
    logical :: in_cloud(a%nz)
    integer :: i, j, lbot, ltop, nz, ns
    nz = a%nz
    ns = size(cloud_layer)
    in_cloud = .false.
    do i = 1, ns
      if (cloud_layer(i) <= 0) cycle
      lbot = cloud_layer(i)
      ltop = lbot
      if (i /= ns) then
        if (cloud_layer(i + 1) < 0) ltop = -cloud_layer(i + 1)
      end if
      do j = max(ltop - 1, 1), lbot
        in_cloud(nz - j + 1) = .true.
      end do
    end do
    if (keep_clear) then
      do i = 1, nz
        if (in_cloud(i)) a%wh(i) = rhcld*saturation_density(tzero/a%t(i))
      end do
      return
    end if
    column = weighted_column(a%wh, nz, (/(.true., i = 1, nz)/))
    if (column == 0.) then
      print *, 'Error in saturate ---  original column water vapor is zero -- can not modify'
      stop
    end if
    do i = 1, nz
      if (in_cloud(i)) a%wh(i) = rhcld*saturation_density(tzero/a%t(i))
    end do
    ! (the reference tags the cloud levels by a negative density, so a clear level counts only if its
    !  density is positive, a cloud level only if it is non-zero, and the top level counts in neither sum)
    clear = weighted_column(a%wh, nz - 1, .not. in_cloud .and. a%wh > 0.)
    cloudy = weighted_column(a%wh, nz - 1, in_cloud .and. a%wh /= 0.)
    if (cloudy == 0) then
      print *, 'Error in saturate --- water vapor density in cloud = 0 ?'
      stop
    end if
    if (clear == 0) then
      cldfac = column/cloudy
      clrfac = 1.e-30
    else
      clrfac = (column - cloudy)/clear
      cldfac = 1.
      if (clrfac < 0) then
        clrfac = 1.e-30
        cldfac = column/cloudy
      end if
    end if
    where (in_cloud .and. a%wh /= 0.)
      a%wh = cldfac*a%wh
    elsewhere
      a%wh = clrfac*a%wh
    end where
    after = 0.
    do i = 1, nz - 1
      after = after + .1*slab_column(a%z(i + 1) - a%z(i), a%wh(i), a%wh(i + 1))
    end do
    a%wh = a%wh*column/after
  contains
    ! sum over levels 1..n of density x the height interval the level stands for (half-way to its neighbours)
    real(kr) function weighted_column(d, n, mask) result(w)
      real(kr), intent(in) :: d(:)
      integer, intent(in) :: n
      logical, intent(in) :: mask(:)
      real(kr) :: zbot, ztop
      integer :: k
      w = 0.
      zbot = a%z(1)
      do k = 1, n
        if (k == 1) then
          ztop = .5*(a%z(2) + zbot)
        else if (k == nz) then
          ztop = a%z(nz)
        else
          ztop = .5*(a%z(k + 1) + a%z(k))
        end if
        if (mask(k)) w = w + .1*(ztop - zbot)*d(k)
        zbot = ztop
      end do
    end function
    real(kr) function saturation_density(x) result(s)
      real(kr), intent(in) :: x
      s = x*exp(18.916758_kr - x*(14.845878_kr + x*2.4918766_kr))
    end function
  end subroutine

  ! relative humidity from temperature (K) and water-vapour density (g/m3) (tauaero.f:1499-1524)
  real(kr) function relative_humidity(t, h2o) result(rh)
    real(kr), intent(in) :: t, h2o
    real(kr) :: a
    a = tzero/t
    rh = h2o/(a*exp(18.916758_kr - a*(14.845878_kr + a*2.4918766_kr)))
  end function

end module sbd_atmos_mod
