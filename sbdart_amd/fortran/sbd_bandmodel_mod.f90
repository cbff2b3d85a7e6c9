! From &INPUT to the per-(wavelength, k-term) work items of the engine: the step in front of the hot
! path (SURVEY 8f row N1; reference drt.f:297-347, 425-533 with rayleigh spectra.f:179-247, solirr
! spectra.f:1367-1415, normom drt.f:1366-1380, depthscl taugas.f:7512-7648).  Covered: gases, Rayleigh,
! clouds and aerosols over a Lambertian surface (constant albedo, the six standard spectra or their mixture); what it does not cover yet is refused
! by name (BRDF surfaces, CKTAU files, ice water in usrcld.dat) and
! still runs from an optics file the reference produced (sbd_io_mod).  Literals: see sbd_tables_mod.
module sbd_bandmodel_mod
  use sbd_grid_mod, only: kr, unset, spectral_grid, nstrms
  use sbd_io_mod, only: optics_t, fatal, warn_file
  use sbd_tables_mod
  use sbd_atmos_mod
  use sbd_gas_mod
  use sbd_cloud_mod
  use sbd_aerosol_mod
  use sbd_filter_mod, only: read_spectrum_file
  use sbd_surface_mod
  use sbd_ckfile_mod
  use omp_lib, only: omp_get_max_threads
  implicit none
  private
  public :: model_input, covered_by_band_model, build_work_items, aerosol_input, gas_depth_report, corint_history, &
            mix_batch, mix_max_terms, assemble_item, expand_work_items
  integer, parameter :: maxmom_all = 299               ! params.f:10

  type model_input                     ! the &INPUT variables this step reads, same names
    integer :: idatm = 4, nf = 2, isalb = 0, kdist = 3, nothrm = -1, ngrid = 0, nstr = 4
    type(aerosol_input) :: aer
    real(kr) :: amix = unset, sza = 0, solfac = 1, albcon = 0, xrsc = 1, zpres = unset, pbar = unset, &
                sclh2o = unset, uw = unset, uo3 = unset, o3trp = unset, ztrp = 0, xgas(11) = unset, xo4 = 1, &
                btemp = unset, ttemp = unset, temis = 0, fisot = 0, phi0 = 0
    real(kr) :: zcloud(ncldz) = 0, tcloud(ncldz) = 0, lwp(ncldz) = 0, nre(ncldz) = 8, rhcld = unset
    integer :: imomc = 3, krhclr = 0
    real(kr) :: zgrid1 = 1, zgrid2 = 30
    real(kr) :: sc(5) = (/1._kr, 0._kr, 0._kr, 0._kr, 0._kr/)   ! ISALB=10: fractions of snow, ocean, sand, vegetation
    logical :: spowder = .false., radiance = .false.
    logical :: corint = .false.            ! with radiances: all 299 phase-function moments go to the engine (drt.f:490-494)
    integer :: numu = 0, nphi = 0
  end type

  ! The run's work items in the COMPACT form of the engine's C ABI (include/sbdart_amd.h, sbd_mix_in): per spectral
  ! point a block lay(:, :, point) -- cloud, aerosol and Rayleigh depths, their scattering depth, then asymmetry factor
  ! and the two factors of every scattering term (slot 1 the cloud when the run has one, then the aerosol_terms) --
  ! and per work item only the gas absorption of its k-term.  The engine forms DTAUC / SSALB / PMOM from these on the
  ! device, statement for statement what depthscl and normom do (taugas.f:7598-7603, drt.f:1390-1395).
  integer, parameter :: mix_max_terms = 6                 ! SBD_MIX_MAX_TERMS
  type mix_batch
    logical :: want = .false.                             ! in: the caller can use the compact form
    logical :: ok = .false.                               ! out: the run fits it, and the arrays below are the run's
    character(len=96) :: why = ''                         ! ... or why not
    integer :: nterm = 0, family(mix_max_terms) = 0
    real(kr), allocatable :: lay(:, :, :)                 ! (nz, 4 + 3 nterm, npoint)
    ! the run's gas model (include/sbdart_amd.h, sbd_gas_model): what gasset / depthscl take besides the wavelength
    logical :: gas_on_device = .false.                    ! in: leave the gas terms to the engine (sbd_fleet_gas_terms):
                                                          ! the band model then delivers no work items, only the points
    logical :: gas_ok = .false.                           ! out: the fields below are set (not with a k-distribution file)
    integer :: kdist = 3
    real(kr) :: amu_gas(2) = 0, xo4 = 1                   ! cosine for the gas terms of the first / the later wavelengths
    real(kr), allocatable :: uu(:, :), z(:)               ! (63, nz) absorber amounts, (nz) altitudes, bottom-up
    real(kr), allocatable :: wl(:)                        ! (npoint)
    ! the scatterers' part on the device as well (include/sbdart_amd.h, sbd_scat_model; sbd_fleet_point_terms): the layer
    ! blocks are then made where the engine reads them and `lay` holds no point (npoint, nch say what it would hold)
    logical :: scat_on_device = .false.                   ! in: the caller wants that (only with gas_on_device)
    logical :: scat_ok = .false.                          ! out: the model below is the run's (sbd_scat.hpp covers the run)
    logical :: scat_dev = .false.                         ! out: ... and the blocks are left to the devices
    integer :: npoint = 0, nch = 0
    type(atmosphere) :: atm                               ! levels bottom-up (Rayleigh)
    type(cloud_deck) :: deck
    type(aerosol_load) :: load
    real(kr) :: xrsc = 1
    integer :: ncloud_term = 0
  end type

contains

  ! DTAUC, SSALB and the moments of one work item from its compact form -- the host-side twin of the engine's
  ! assemble_kernel, for the rare paths that want the numbers on the host (CHEKIN's report on a bad item)
  subroutine assemble_item(mix, ipoint, dtaug, nmom, dtau, wreal, pmom)
    type(mix_batch), intent(in) :: mix
    integer, intent(in) :: ipoint, nmom
    real(kr), intent(in) :: dtaug(:)
    real(kr), intent(out) :: dtau(:), wreal(:), pmom(0:, :)
    real(kr) :: q, pk, dtsct
    integer :: l, k, t
    do l = 1, size(dtaug)
      dtau(l) = dtaug(l) + mix%lay(l, 1, ipoint) + mix%lay(l, 2, ipoint) + mix%lay(l, 3, ipoint)
      dtsct = mix%lay(l, 4, ipoint)
      wreal(l) = 0.
      if (dtau(l) > tiny(1._kr)) wreal(l) = dtsct/dtau(l)
      pmom(0, l) = 1.
      do k = 1, nmom
        q = 0.
        do t = 1, mix%nterm
          pk = 0.
          if (mix%family(t) == 3) pk = mix%lay(l, 2 + 3*t, ipoint)**k
          if (mix%family(t) == 2 .and. k == 2) pk = .1
          q = q + pk*mix%lay(l, 3 + 3*t, ipoint)*mix%lay(l, 4 + 3*t, ipoint)
        end do
        if (k == 2) q = q + .1*mix%lay(l, 3, ipoint)
        if (dtsct /= 0.) q = q/dtsct
        pmom(k, l) = q
      end do
    end do
  end subroutine

  ! Device-side gas terms (mix_batch%gas_on_device): the band model delivered ONE record per spectral point; with the
  ! number of k-terms nk(point) and their weights wt(:, point) back from the engine (sbd_fleet_gas_terms) the records
  ! become the run's work items -- a point's record once per k-term, in the order the reference's kd_loop makes them
  subroutine expand_work_items(recs, nrec, nk, wt)
    type(optics_t), allocatable, intent(inout) :: recs(:)
    integer, intent(inout) :: nrec
    integer, intent(in) :: nk(:)
    real(kr), intent(in) :: wt(:, :)
    type(optics_t), allocatable :: items(:)
    integer, allocatable :: first(:)
    integer :: i, k, n, nthreads
    allocate(first(nrec))
    n = 0
    do i = 1, nrec
      first(i) = n
      n = n + nk(i)
    end do
    allocate(items(n))
    nthreads = max(1, min(omp_get_max_threads(), nrec/64, 16))
    !$omp parallel do schedule(static) num_threads(nthreads) private(k)
    do i = 1, nrec
      do k = 1, nk(i)
        items(first(i) + k) = recs(i)
        items(first(i) + k)%kd = k; items(first(i) + k)%nk = nk(i); items(first(i) + k)%wt = wt(k, i)
      end do
    end do
    !$omp end parallel do
    call move_alloc(items, recs)
    nrec = n
  end subroutine

  ! .true. when every switch of the run is inside the first slice; otherwise why not
  logical function covered_by_band_model(m, why) result(ok)
    type(model_input), intent(in) :: m
    character(len=*), intent(out) :: why
    why = ''
    if (m%kdist < -1) why = 'k-distribution mode (kdist < -1)'
    if (m%nf == -2 .and. m%kdist /= -1) why = 'solar spectrum from a k-distribution file (nf=-2) without kdist=-1'
    ok = len_trim(why) == 0
  end function

  ! Rayleigh optical depth of every layer (1 = top, which is the column above the highest level taken with
  ! a 5 km scale height); density-weighted slabs between the levels (spectra.f:179-247)
  subroutine rayleigh_depths(wl, a, dtaur)
    real(kr), intent(in) :: wl
    type(atmosphere), intent(in) :: a
    real(kr), intent(out) :: dtaur(a%nz)
    real(kr), parameter :: fit1 = 9.38076e+18, fit2 = -1.08426e+09
    real(kr) :: v, sig, lower, upper, dz
    integer :: i, lev, nz
    nz = a%nz
    v = 10000./wl
    sig = v**4/(fit1 + fit2*v**2)
    dtaur(1) = sig*(a%p(nz)/pzero)/(a%t(nz)/tzero)*5.
    do i = 2, nz
      lev = nz - i + 1
      lower = (a%p(lev)/pzero)/(a%t(lev)/tzero)
      upper = (a%p(lev + 1)/pzero)/(a%t(lev + 1)/tzero)
      dz = a%z(lev + 1) - a%z(lev)
      if (lower == upper) then
        dtaur(i) = .5*sig*dz*(lower + upper)
      else
        dtaur(i) = sig*dz*(upper - lower)/log(upper/lower)
      end if
    end do
  end subroutine

  ! extraterrestrial solar irradiance (W/m2/um) at wl, linear in the spectrum's own grid (spectra.f:1367-1415)
  real(kr) function solar_irradiance(wl, nf, w, s) result(e)
    real(kr), intent(in) :: wl, w(:), s(:)               ! the spectrum (solar_spectrum): wavelengths, irradiance
    integer, intent(in) :: nf
    real(kr) :: wt
    integer :: j
    if (nf == 0) then
      e = 1.
      return
    end if
    j = bracket(w, wl)
    wt = (wl - w(j))/(w(j + 1) - w(j))
    wt = max(0._kr, min(1._kr, wt))
    e = s(j)*(1. - wt) + s(j + 1)*wt
  end function

  ! NF: 1 5S, 2 LOWTRAN7, 3 MODTRAN3 (tables), -1 the file solar.dat, 0 none (unit irradiance)
  subroutine solar_spectrum(nf, w, s)
    integer, intent(in) :: nf
    real(kr), allocatable, intent(out) :: w(:), s(:)
    character(len=4) :: name
    select case (nf)
    case (-1)
      call read_spectrum_file('solar.dat', 5000, w, s)
    case (1:3)
      write(name, '(a,i1)') 'sun', nf
      w = tbl(name//'.wl'); s = tbl(name//'.irr')
    case default
      w = (/0._kr, 1._kr/); s = (/1._kr, 1._kr/)
    end select
  end subroutine

  ! albedo spectrum of the surface: ISALB 0 constant, 1-6 snow / clear water / lake water / sea water /
  ! sand / vegetation, 10 a mixture of snow, sea water, sand and vegetation (suralb, spectra.f:61-118)
  subroutine surface_spectrum(isalb, albcon, sc, wlalb, alb)
    integer, intent(in) :: isalb
    real(kr), intent(in) :: albcon, sc(5)
    real(kr), allocatable, intent(out) :: wlalb(:), alb(:)
    character(len=4) :: name
    select case (isalb)
    case (-1)
      call read_spectrum_file('albedo.dat', 5000, wlalb, alb)
    case (0)
      wlalb = (/0._kr, huge(0._kr)/)
      alb = (/albcon, albcon/)
    case (1:6)
      write(name, '(a,i1)') 'alb', isalb
      wlalb = tbl(name//'.wl'); alb = tbl(name//'.r')
    case (10)
      wlalb = tbl('alb6.wl')
      alb = tbl('alb1.r')*sc(1)
      alb = tbl('alb4.r')*sc(2) + alb
      alb = tbl('alb5.r')*sc(3) + alb
      alb = tbl('alb6.r')*sc(4) + alb
    end select
  end subroutine

  ! albedo at wl, linear in the spectrum's grid; outside it the end value, with the reference's warning 18
  real(kr) function surface_albedo(wlalb, alb, wl) result(r)
    real(kr), intent(in) :: wlalb(:), alb(:), wl
    character(len=9) :: num
    real(kr) :: wt
    integer :: j, n
    n = size(wlalb)
    if (wl < wlalb(1)) then
      write(num, '(f9.3)') wlalb(1)
      call warn_file(18, 'SALBEDO--spectral range error, wlinf lt '//num)
    end if
    if (wl > wlalb(n)) then
      write(num, '(f9.3)') wlalb(n)
      call warn_file(18, 'SALBEDO--spectral range error, wlsup gt '//num)
    end if
    j = bracket(wlalb, wl)
    wt = (wl - wlalb(j))/(wlalb(j + 1) - wlalb(j))
    wt = max(0._kr, min(1._kr, wt))
    r = alb(j)*(1. - wt) + alb(j + 1)*wt
  end function

  ! weight that fades the slant-path correction out: with wavelength across 3.9-4.1 um (thermal emission
  ! takes over from the sun) and with scattering optical depth above 1 (taugas.f:7625-7647)
  pure real(kr) function correction_weight(wl, tsc) result(ramp)
    real(kr), intent(in) :: wl, tsc
    real(kr), parameter :: wllo = 3.9, wlhi = 4.1
    ramp = (wlhi - wl)/(wlhi - wllo)
    ramp = max(min(1._kr, ramp), 0._kr)
    ramp = ramp*exp(1. - max(tsc, 1._kr))
  end function

  ! IOUT = 2: no radiative transfer, one line per wavelength with the gas optical depth of the slant path to the
  ! surface by absorber (water lines + continuum, CO2, O3, N2O, CO, CH4, O2 + N2, trace gases, total)
  ! (drt.f:440-445 with taugas' diagnostic print, taugas.f:2497-2500).  An absorber without a band at a
  ! wavelength shows the value of the last wavelength where it had one, as in the reference.
  subroutine gas_depth_report(m, grid)
    type(model_input), intent(in) :: m
    type(spectral_grid), intent(in) :: grid
    type(atmosphere) :: atm
    type(trace_gases) :: mix
    type(cloud_deck) :: deck
    type(gas_spectrum) :: spec
    real(kr), allocatable :: uu(:, :), dc(:), dl(:)
    real(kr) :: pbar, amu0, wl, wvlo, wvhi, col(9), carry(nmol)
    real(kr), parameter :: dtor = 3.1415926536_kr/180.
    integer :: nz, iwl
    carry = 0.
    if (m%idatm == 0) then
      atm = user_atmosphere()
    else
      atm = model_atmosphere(m%idatm)
    end if
    if (m%amix > -1.) call mix_in(atm, m%amix)
    if (m%ngrid /= 0) call regrid(atm, m%zgrid1, m%zgrid2, m%ngrid)
    pbar = m%pbar
    if (m%zpres /= unset) pbar = pressure_at(atm, m%zpres)
    call rescale_profiles(atm, m%sclh2o, m%uw, m%uo3, m%o3trp, m%ztrp, pbar)
    call set_trace_gases(mix, m%xgas, m%xo4)
    nz = atm%nz
    deck = new_cloud_deck(atm%z, m%zcloud, m%tcloud, m%lwp, m%nre, m%imomc)
    if (m%rhcld >= 0) call saturate_clouds(atm, deck%layer, m%rhcld, m%krhclr == 1)
    allocate(uu(mxq, nz), dc(nz), dl(nz))
    call absorber_columns(atm, mix, uu)
    call gas_tables_init()
    amu0 = cos(m%sza*dtor)
    print *, 'nwl', grid%n
    do iwl = 1, grid%n
      call grid%band(iwl - 1, wl, wvlo, wvhi)
      spec = spectrum_at(wl, mix%xo4)
      call path_depths(spec, uu, amu0, atm%z, nz, dc, dl, col, carry)
      print '(f11.4,9es11.3)', wl, col(1) + col(2), col(3:9), sum(col)
    end do
  end subroutine

  ! DISORT switches its CORINT argument off IN PLACE when a call has no beam or no scattering (disort.f:2695-2696)
  ! and the reference passes its namelist variable (drt.f:541): every LATER call of the run goes without the
  ! intensity corrections, and from the next wavelength on only NSTR+2 moments are prepared (drt.f:490-494).
  ! The items are walked in call order (those the sensor filter removes are not calls, drt.f:461-462); flag 16
  ! of an item = corrections still requested when its call is made.
  subroutine corint_history(recs, nrec, ssalb)
    type(optics_t), intent(inout) :: recs(:)
    integer, intent(in) :: nrec
    real(kr), intent(in) :: ssalb(:, :)
    logical :: live
    integer :: i, full, nm
    if (nrec == 0) return
    if (iand(recs(1)%flags, 16) == 0) return
    live = .true.
    full = recs(1)%nmom; nm = full
    do i = 1, nrec
      if (i > 1) then
        if (recs(i)%iwl /= recs(i - 1)%iwl) nm = merge(full, min(recs(i)%nstr + 2, nstrms), live)
      end if
      recs(i)%nmom = nm
      if (.not. live) recs(i)%flags = recs(i)%flags - iand(recs(i)%flags, 16)
      if (recs(i)%ff /= 0._kr .and. (recs(i)%fbeam == 0._kr .or. sum(ssalb(:, i)) == 0._kr)) live = .false.
    end do
  end subroutine

  ! The work items of a run, ordered by wavelength then k-term: the scalars of every item in recs (no arrays
  ! allocated there), the layer arrays in contiguous batch arrays -- the engine's input layout.  The wavelengths
  ! are independent of each other (the reference's saved state is replaced by values prepared once per run), so
  ! the loop over them is an OpenMP parallel loop: every wavelength fills its own MK slots, a second parallel
  ! loop closes the slots up.  No allocation inside the loops.
  subroutine build_work_items(m, grid, umu, phi, recs, nrec, atm, bdtauc, bssalb, bpmom, btemper, ck, mixb)
    type(model_input), intent(in) :: m
    type(spectral_grid), intent(in) :: grid
    type(ck_file), intent(in), optional :: ck           ! KDIST = -1: the k-distribution file pair; `grid` is its band list
    ! mixb%want: hand the items over in compact form when the run fits it (mixb%ok): bdtauc then holds the GAS depth of
    ! every item, bssalb and bpmom stay empty, mixb%lay holds the spectral points' blocks
    type(mix_batch), intent(inout), optional :: mixb
    real(kr), intent(in) :: umu(:), phi(:)
    type(optics_t), allocatable, intent(out) :: recs(:)
    integer, intent(out) :: nrec
    type(atmosphere), intent(out) :: atm
    real(kr), allocatable, intent(out) :: bdtauc(:, :), bssalb(:, :), bpmom(:, :, :), btemper(:)
    type(trace_gases) :: mix
    type(cloud_deck) :: deck
    type(aerosol_load) :: load
    type(layer_clouds) :: lcloud
    real(kr), allocatable :: uu(:, :), temper(:), wlalb(:), alb(:), wsun(:), ssun(:)
    real(kr), allocatable :: sd(:, :), ss(:, :), swt(:, :), swl(:), slo(:), shi(:), sfb(:), salb(:), sbit(:, :)
    type(surface_model) :: surf
    logical, allocatable :: splank(:)
    integer, allocatable :: nk_of(:), first(:)
    real(kr) :: pbar, amu0, btemp, ttemp, rh_surface, wv1, wv2
    real(kr), allocatable :: run_wl(:)
    real(kr), parameter :: dtor = 3.1415926536_kr/180.
    integer :: nz, nmom, iwl, kd, i
    integer :: nthreads, mkt
    logical :: from_ck, compact, aer_ok, gas_dev, scat_dev
    integer :: ncloud_term, naer_term, aer_family(mix_max_terms), nch
    integer(kind=8) :: tk(4), tkrate
    character(len=8) :: tenv
    integer :: tlen, tstat

    ! ---- once per run: profiles, rescaling, absorber amounts, clouds, aerosols, surface (drt.f:297-423) ----
    call system_clock(tk(1), tkrate)
    from_ck = m%kdist == -1
    if (from_ck .and. .not. present(ck)) call fatal('kdist=-1: no k-distribution file was read')
    mkt = mk                                             ! k-term slots per spectral point
    if (from_ck) then
      ! levels, pressures and temperatures from CKATM; no gas amounts are needed (gasinit, taugas.f:7297-7390)
      mkt = ck%maxk
      atm%nz = ck%nz
      atm%z = ck%z; atm%p = ck%p; atm%t = ck%t
      allocate(atm%wh(ck%nz), atm%wo(ck%nz))
      atm%wh = 0; atm%wo = 0
      nz = atm%nz
      rh_surface = relative_humidity(atm%t(1), ck%h2oden)
    else
      if (m%idatm == 0) then
        atm = user_atmosphere()
      else
        atm = model_atmosphere(m%idatm)
      end if
      if (m%amix > -1.) call mix_in(atm, m%amix)
      if (m%ngrid /= 0) call regrid(atm, m%zgrid1, m%zgrid2, m%ngrid)
      nz = atm%nz
      pbar = m%pbar
      if (m%zpres /= unset) pbar = pressure_at(atm, m%zpres)
      call rescale_profiles(atm, m%sclh2o, m%uw, m%uo3, m%o3trp, m%ztrp, pbar)
      call set_trace_gases(mix, m%xgas, m%xo4)
      rh_surface = relative_humidity(atm%t(1), atm%wh(1))
    end if
    ! level temperatures top-down and the default boundary temperatures (drt.f:330-335) -- taken BEFORE the
    ! sub-surface layer is added, as the reference does (the extra bottom level keeps temperature zero)
    allocate(temper(0:nz + merge(1, 0, m%spowder)))
    temper = 0.
    temper(0) = atm%t(nz)
    do i = 1, nz
      temper(i) = atm%t(nz + 1 - i)
    end do
    btemp = m%btemp; if (btemp < 0.) btemp = temper(nz)
    ttemp = m%ttemp; if (ttemp < 0.) ttemp = temper(0)
    ! SPOWDER: one more layer under the surface, 1 km thick, without gases or Rayleigh scattering; what
    ! scatters in it is the cloud the user puts there (drt.f:337-347, taugas.f:7557-7560)
    if (m%spowder) then
      if (nz >= 65) then
        print *, 'Error --- nz < mxly is required with spowder option'
        stop
      end if
      atm%z = (/-1._kr, atm%z/)
      atm%p = (/1.1*atm%p(1), atm%p/)
      atm%t = (/btemp, atm%t/)
      atm%wh = (/0._kr, atm%wh/)
      atm%wo = (/0._kr, atm%wo/)
      nz = nz + 1
      atm%nz = nz
    end if
    deck = new_cloud_deck(atm%z, m%zcloud, m%tcloud, m%lwp, m%nre, m%imomc)
    if (m%rhcld >= 0 .and. .not. from_ck) call saturate_clouds(atm, deck%layer, m%rhcld, m%krhclr == 1)   ! drt.f:357-365
    allocate(uu(mxq, nz))
    if (.not. from_ck) call absorber_columns(atm, mix, uu)
    nmom = min(m%nstr + 2, nstrms)                          ! (two more than NSTR: room for the NSTR retry)
    if (m%radiance .and. m%corint) nmom = maxmom_all
    amu0 = cos(m%sza*dtor)
    surf = new_surface_model(m%isalb, m%sc)                 ! (ISALB 7, 8, 9: a bidirectional surface, no albedo spectrum)
    if (surf%ibdrf == 0) call surface_spectrum(m%isalb, m%albcon, m%sc, wlalb, alb)
    call solar_spectrum(m%nf, wsun, ssun)
    if (.not. from_ck) call gas_tables_init()
    call cloud_tables_init()
    if (deck%nslot == 0 .and. m%nre(1) == 0.) lcloud = read_layer_clouds(nz)      ! drt.f:501-502
    load = new_aerosol_load(m%aer, atm%z, rh_surface)
    if (m%aer%iaer == -1) then                              ! aerosol.dat is read in wavelength order: walk it first
      allocate(run_wl(grid%n))
      do iwl = 1, grid%n
        call grid%band(iwl - 1, run_wl(iwl), wv1, wv2)
      end do
      call plan_aerosol_file(load, run_wl)
    end if

    ! ---- does the run fit the compact form?  (every scatterer one term GETMOM(family, g) x two factors) ----
    compact = .false.; gas_dev = .false.; scat_dev = .false.
    if (present(mixb)) then
      if (.not. mixb%want) mixb%gas_on_device = .false.
      mixb%scat_dev = .false.
    end if
    ncloud_term = 0; naer_term = 0; nch = 4
    call aerosol_terms(load, naer_term, aer_family, aer_ok)      ! (also sizes the term recorder of every wavelength)
    if (deck%nslot > 0 .or. lcloud%given) ncloud_term = 1
    if (present(mixb)) then
      mixb%ok = .false.; mixb%why = ''
      if (mixb%want) then
        if (m%radiance .and. m%corint) then
          mixb%why = 'intensity corrections (299 moments, CORINT history on the host)'
        else if (m%spowder) then
          mixb%why = 'SPOWDER (normom and depthscl see different Rayleigh depths in the sub-surface layer)'
        else if (surf%ibdrf == 1 .and. .not. surf%as_albedo) then
          mixb%why = 'ocean surface (per-item constants)'
        else if (ncloud_term == 1 .and. (m%imomc < 1 .or. m%imomc > 3)) then
          mixb%why = 'tabulated cloud phase function (imomc)'
        else if (deck%nslot > 0 .and. .not. one_cloud_per_layer(deck, nz)) then
          mixb%why = 'two clouds in one layer'
        else if (.not. aer_ok) then
          mixb%why = 'aerosol moments that are not a function of one asymmetry factor'
        else if (ncloud_term + naer_term > mix_max_terms) then
          mixb%why = 'more scattering terms per layer than the compact form holds'
        else
          compact = .true.
          mixb%ok = .true.
          mixb%nterm = ncloud_term + naer_term
          mixb%family = 0
          if (ncloud_term == 1) mixb%family(1) = m%imomc
          mixb%family(ncloud_term + 1:ncloud_term + naer_term) = aer_family(1:naer_term)
          nch = 4 + 3*mixb%nterm
          mixb%gas_ok = .not. from_ck
          gas_dev = mixb%gas_ok .and. mixb%gas_on_device
          mixb%gas_on_device = gas_dev
          ! the scatterers on the device too: what sbd_scat.hpp covers -- a cloud deck (not usrcld.dat), IAER 1..5
          mixb%scat_ok = .not. lcloud%given .and. load%iaer /= -1
          if (mixb%scat_ok) then
            mixb%atm = atm; mixb%deck = deck; mixb%load = load; mixb%xrsc = m%xrsc; mixb%ncloud_term = ncloud_term
          end if
          scat_dev = gas_dev .and. mixb%scat_on_device .and. mixb%scat_ok
          mixb%scat_dev = scat_dev
          mixb%npoint = grid%n; mixb%nch = nch
          if (allocated(mixb%lay)) deallocate(mixb%lay)
          allocate(mixb%lay(nz, nch, merge(0, grid%n, scat_dev)))
          if (mixb%gas_ok) then
            mixb%kdist = m%kdist; mixb%xo4 = mix%xo4
            mixb%amu_gas = amu0
            if (m%sza >= 90.) mixb%amu_gas(2) = 1.       ! (drt.f:433-455: the cosine is reset after the first gasset)
            mixb%uu = uu; mixb%z = atm%z
            if (allocated(mixb%wl)) deallocate(mixb%wl)
            allocate(mixb%wl(grid%n))
          end if
        end if
      end if
    end if

    if (present(mixb)) then
      if (.not. compact) mixb%gas_on_device = .false.
    end if
    ! (the phase-function moments belong to the WAVELENGTH: one block per spectral point, shared by its k-terms --
    !  drt.f:476-533 computes them before the k loop -- and handed to the engine that way, sbd_batch_in%pmom_row)
    if (compact) then
      allocate(bpmom(0:0, 1, 1))
    else
      allocate(bpmom(0:nmom, nz, grid%n))
    end if
    allocate(nk_of(grid%n), first(grid%n), sd(nz, merge(1, mkt*grid%n, gas_dev)), ss(nz, merge(1, mkt*grid%n, compact)), &
             swt(mkt, grid%n), swl(grid%n), slo(grid%n), shi(grid%n), sfb(grid%n), salb(grid%n), splank(grid%n), &
             sbit(4, grid%n))
    sbit = 0
    ! threads: one per 64 wavelengths, at most 16 (measured on the 256-core GPU box, 75 001 wavelengths: 1.2 s
    ! with 1 thread, 0.25-0.31 s with 16, 0.46-0.5 s with 64: first-touch page faults of the 2.5 GB of slot
    ! and batch arrays, not arithmetic, set the pace beyond that)
    nthreads = max(1, min(omp_get_max_threads(), grid%n/64, 16))
    call system_clock(tk(2))
    !$omp parallel do schedule(dynamic, 16) num_threads(nthreads)
    do iwl = 1, grid%n
      call one_wavelength(iwl)
    end do
    !$omp end parallel do
    call system_clock(tk(3))
    nrec = 0
    do iwl = 1, grid%n
      first(iwl) = nrec + 1
      nrec = nrec + nk_of(iwl)
    end do
    allocate(recs(nrec), bdtauc(nz, merge(1, nrec, gas_dev)), bssalb(nz, merge(1, nrec, compact)), btemper(0:nz))
    btemper = temper
    !$omp parallel do schedule(static) num_threads(nthreads) private(kd, i)
    do iwl = 1, grid%n
      do kd = 1, nk_of(iwl)
        i = first(iwl) + kd - 1
        if (.not. gas_dev) bdtauc(:, i) = sd(:, mkt*(iwl - 1) + kd)
        if (.not. compact) bssalb(:, i) = ss(:, mkt*(iwl - 1) + kd)
        recs(i)%nlyr = nz; recs(i)%nstr = m%nstr; recs(i)%nmom = nmom; recs(i)%numu = size(umu); recs(i)%nphi = size(phi)
        recs(i)%flags = merge(1, 0, splank(iwl)) + merge(0, 2, m%radiance) + merge(16, 0, m%radiance .and. m%corint)
        recs(i)%kd = kd; recs(i)%nk = nk_of(iwl); recs(i)%iwl = iwl
        recs(i)%wl = swl(iwl); recs(i)%wt = swt(kd, iwl); recs(i)%ff = 1.
        recs(i)%wvnmlo = slo(iwl); recs(i)%wvnmhi = shi(iwl); recs(i)%fbeam = sfb(iwl)
        recs(i)%umu0 = merge(1._kr, amu0, m%sza >= 90.); recs(i)%phi0 = m%phi0; recs(i)%albedo = salb(iwl)
        recs(i)%btemp = btemp; recs(i)%ttemp = ttemp; recs(i)%temis = m%temis; recs(i)%fisot = m%fisot
        if (.not. surf%as_albedo) then
          recs(i)%ibdrf = surf%ibdrf; recs(i)%bpar = surf%par; recs(i)%bitem = sbit(:, iwl)
        end if
        if (from_ck) then
          recs(i)%ib = ck%ib(iwl); recs(i)%nb = ck%nb(iwl); recs(i)%ewcoef = ck%ewcoef(iwl)
        end if
      end do
    end do
    !$omp end parallel do
    call system_clock(tk(4))
    call get_environment_variable('SBD_TIMING', tenv, tlen, tstat)
    if (tstat == 0 .and. tlen > 0) write(0, '(a,3(f8.4,a))') 'sbdart_amd: band model: per-run setup ', &
      real(tk(2) - tk(1), 8)/real(tkrate, 8), ' s, wavelength loop ', real(tk(3) - tk(2), 8)/real(tkrate, 8), &
      ' s, records ', real(tk(4) - tk(3), 8)/real(tkrate, 8), ' s'

  contains

    subroutine one_wavelength(iw)
      integer, intent(in) :: iw
      type(gas_spectrum) :: spec
      real(kr) :: dtaur(nz), dtauk(nz, 2*max(mk, mkt)), dtaugc(nz), dtaug(nz), scat(nz), dtauc(nz), wcld(nz), &
                  pmom(0:merge(2, nmom, compact), nz), dtaua(nz), waer(nz)
      real(kr) :: trm_c(nz, 3), trm_a(nz, 3, naerz + 1)      ! (boundary layer + every stratospheric layer)
      integer :: nmw
      real(kr) :: wl, wvlo, wvhi, dwl, flxin, rsfc, gwk(max(mk, mkt)), wt, tsc, tglv, tgls, afac, ramp, amu_gas, amu_sun
      integer :: nk, k, l
      logical :: plank
      call grid%band(iw - 1, wl, wvlo, wvhi)
      ! no sun (SZA >= 90): the reference hands DISORT a unit cosine and, from the second wavelength on, also
      ! evaluates the gas terms for a vertical path (drt.f:433-455: the cosine is reset after the first gasset)
      amu_gas = amu0; amu_sun = amu0
      if (m%sza >= 90.) then
        amu_sun = 1.
        if (iw > 1) amu_gas = 1.
      end if
      if (gas_dev) then                                          ! (the engine evaluates the gas terms: sbd_fleet_gas_terms)
        nk = 1
        gwk = 0.; gwk(1) = 1.
      else if (from_ck) then                                     ! readk: weights and depths of this sub-band's k-terms
        nk = ck%nk(iw)
        gwk(1:nk) = ck%gwk(1:nk, iw)
        dtauk(:, 1:nk) = ck%dtauk(:, 1:nk, iw)
        dtaugc = 0.
      else
        spec = spectrum_at(wl, mix%xo4)
        call gas_terms(m%kdist, spec, uu, amu_gas, atm%z, nz, nk, gwk, dtauk, dtaugc)
      end if
      dwl = 10000./wvlo - 10000./wvhi
      if (m%nf == -2) then                                       ! the file's own extra-terrestrial flux (drt.f:446-448)
        flxin = ck%etirr(iw)*m%solfac
      else
        flxin = solar_irradiance(wl, m%nf, wsun, ssun)*dwl*m%solfac
      end if
      if (m%nf == 0) flxin = dwl
      if (m%sza >= 90.) flxin = 0.
      if (m%nothrm < 0) then
        plank = wl > 2.
      else
        plank = m%nothrm == 0
      end if
      if (surf%ibdrf /= 0) then
        rsfc = 0.                                                 ! (LAMBER off: DISORT never reads ALBEDO)
        ! the ocean's water constants at BDREF's wavelength, the middle of the band in wavenumber (spectra.f:284)
        if (surf%ibdrf == 1) call ocean_constants(surf, 20000./(wvhi + wvlo), sbit(1, iw), sbit(2, iw), sbit(3, iw))
        if (surf%as_albedo) then                                  ! ISALB -7, -8, -9 (drt.f:478-484): DREF at cos(SZA) as it
          rsfc = flux_albedo(surf, sbit(:, iw), amu0)             ! is, also below the horizon; DREF warns outside [0,1]
          if (rsfc < 0._kr .or. rsfc > 1._kr) then                ! (disort.f:5279-5280), the driver clamps
            !$omp critical (sbd_surface_warning)
            call warn_file(8, 'DREF--albedo value not in (0,1)')
            !$omp end critical (sbd_surface_warning)
          end if
          rsfc = max(0._kr, min(rsfc, 1._kr))
          sbit(:, iw) = 0.
        end if
      else if (wl < wlalb(1) .or. wl > wlalb(size(wlalb))) then   ! (writes the reference's warning file: one at a time)
        !$omp critical (sbd_surface_warning)
        rsfc = max(0._kr, min(surface_albedo(wlalb, alb, wl), 1._kr))
        !$omp end critical (sbd_surface_warning)
      else
        rsfc = max(0._kr, min(surface_albedo(wlalb, alb, wl), 1._kr))
      end if
      if (scat_dev) then                                         ! (the engine makes the layer blocks: sbd_fleet_point_terms)
        if (mixb%gas_ok) mixb%wl(iw) = wl
        nk_of(iw) = nk
        swl(iw) = wl; slo(iw) = wvlo; shi(iw) = wvhi; sfb(iw) = flxin; salb(iw) = rsfc; splank(iw) = plank
        swt(1, iw) = 1.
        return
      end if
      call rayleigh_depths(wl, atm, dtaur)
      if (m%xrsc /= 1._kr) dtaur = m%xrsc*dtaur
      ! clouds and aerosols, then the phase-function moments of the scattering mixture: every scatterer adds
      ! moment x scattering depth, Rayleigh 0.1 in the second moment; normalised by the total (drt.f:1366-1380)
      pmom = 0.
      dtauc = 0.; wcld = 0.
      ! (compact form: the scatterers leave as terms -- asymmetry factor and factors -- and no moment is formed here:
      !  the routines run with two moments, GETMOM's Rayleigh family writes the second)
      nmw = merge(2, nmom, compact)
      trm_c = 0.
      if (deck%nslot > 0) then
        call cloud_depths(deck, wl, nz, nmw, dtauc, wcld, pmom, trm_c)
      else if (lcloud%given) then
        call layer_cloud_depths(lcloud, m%imomc, wl, nz, nmw, dtauc, wcld, pmom, trm_c)
      end if
      call aerosol_depths(load, wl, nz, nmw, dtaua, waer, pmom, iw, trm_a)
      do l = 1, nz
        scat(l) = dtauc(l)*wcld(l) + dtaua(l)*waer(l) + dtaur(l)
      end do
      if (compact) then
        if (mixb%gas_ok) mixb%wl(iw) = wl
        mixb%lay(:, 1, iw) = dtauc; mixb%lay(:, 2, iw) = dtaua; mixb%lay(:, 3, iw) = dtaur; mixb%lay(:, 4, iw) = scat
        if (ncloud_term == 1) mixb%lay(:, 5:7, iw) = trm_c
        do k = 1, naer_term
          mixb%lay(:, 5 + 3*(ncloud_term + k - 1):7 + 3*(ncloud_term + k - 1), iw) = trm_a(:, :, k)
        end do
      else
        do l = 1, nz
          pmom(2, l) = pmom(2, l) + .1*dtaur(l)
          if (scat(l) /= 0.) pmom(:, l) = pmom(:, l)/scat(l)
        end do
        pmom(0, :) = 1.
      end if

      nk_of(iw) = nk
      swl(iw) = wl; slo(iw) = wvlo; shi(iw) = wvhi; sfb(iw) = flxin; salb(iw) = rsfc; splank(iw) = plank
      if (m%spowder) dtaur(nz) = 0.                                  ! (depthscl does this at every k-term)
      if (gas_dev) then                                              ! one record per point for now: expand_work_items
        swt(1, iw) = 1.
        return
      end if
      do k = 1, nk
        ! ---- gas depth of this k-term with the slant-path correction policy KDIST (depthscl) ----
        wt = gwk(k)
        if (from_ck) then                                          ! depths straight from the file (taugas.f:7564-7566)
          dtaug = dtauk(:, k)
        else if (m%kdist == 0 .or. nk == 1) then
          wt = 1.
          tsc = 0.; tglv = 0.; tgls = 0.
          do l = 1, nz
            tglv = tglv + dtauk(l, 1)
            tgls = tgls + dtauk(l, 1 + mk)
            tsc = tsc + dtaur(l) + dtauc(l) + dtaua(l)
            afac = 1.
            if (tglv > .001) afac = tgls/tglv
            ramp = correction_weight(wl, tsc)
            afac = afac*ramp + 1. - ramp
            dtaug(l) = dtaugc(l) + dtauk(l, 1)*afac
          end do
        else if (m%kdist == 1) then
          dtaug = dtaugc + dtauk(:, k)
        else if (m%kdist == 2) then
          dtaug = dtaugc + dtauk(:, k + mk)
        else
          tsc = 0.
          do l = 1, nz
            tsc = tsc + dtaur(l) + dtauc(l) + dtaua(l)
            ramp = correction_weight(wl, tsc)
            dtaug(l) = dtaugc(l) + dtauk(l, k)*(1. - ramp) + dtauk(l, k + mk)*ramp
          end do
        end if
        if (m%spowder) dtaug(nz) = 0.
        ! ---- the work item's layer arrays ----
        swt(k, iw) = wt
        if (compact) then
          sd(:, mkt*(iw - 1) + k) = dtaug                       ! (the gas alone: the device adds the point's scatterers)
          cycle
        end if
        if (k == 1) bpmom(:, :, iw) = pmom
        do l = 1, nz
          sd(l, mkt*(iw - 1) + k) = dtaug(l) + dtauc(l) + dtaua(l) + dtaur(l)
          if (sd(l, mkt*(iw - 1) + k) > tiny(1._kr)) then
            ss(l, mkt*(iw - 1) + k) = (dtauc(l)*wcld(l) + dtaua(l)*waer(l) + dtaur(l))/sd(l, mkt*(iw - 1) + k)
          else
            ss(l, mkt*(iw - 1) + k) = 0.
          end if
        end do
      end do
    end subroutine
  end subroutine

end module sbd_bandmodel_mod
