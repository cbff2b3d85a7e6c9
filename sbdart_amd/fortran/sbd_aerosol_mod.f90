! Aerosols per wavelength (reference: tauaero.f -- tauaero 1175-1359, denprfl 1361-1447, aervint
! 1449-1497, aeroden 1134-1173, module procedures aerzstd 88-140, aerbwi 177-251, aestrat 253-403,
! usraer 405-594, stdaer 596-1130).  Part of SURVEY 8f row N1.
!   boundary layer:  IAER 1-4 rural / urban / oceanic / tropospheric at relative humidity RHAER,
!                    IAER 5 a user spectrum (WLBAER, QBAER, WBAER, GBAER, ABAER); amount from the
!                    visibility VIS or the 0.55 um optical depth TBAER; vertical profile standard or
!                    ZBAER / DBAER
!   stratosphere:    up to five layers ZAER with 0.55 um optical depth TAERST of model JAER 1-4
!   aerosol.dat:    IAER -1, per-layer optical depth, single-scattering albedo and moments at the file's
!                    wavelengths (aeread 1526-1713)
! An aerosol_load is prepared once per run; aerosol_depths evaluates it at a wavelength.  Literals: see
! sbd_tables_mod.
module sbd_aerosol_mod
  use sbd_grid_mod, only: kr, unset
  use sbd_tables_mod
  use sbd_atmos_mod, only: bracket
  use sbd_cloud_mod, only: phase_moments, layers_of_altitudes
  implicit none
  private
  public :: aerosol_input, aerosol_load, new_aerosol_load, aerosol_depths, naerz, naerb, plan_aerosol_file, aerosol_terms

  integer, parameter :: naerz = 5, naerb = 150, naerw = 47
  real(kr), parameter :: wl55 = 0.55

  type aerosol_input                     ! the &INPUT variables, same names
    integer :: iaer = 0, jaer(naerz) = 0, imoma = 3, nosct = 0
    real(kr) :: zaer(naerz) = 0, taerst(naerz) = 0, vis = unset, tbaer = unset, abaer = 0, rhaer = unset
    real(kr) :: wlbaer(naerb) = unset, qbaer(naerb) = unset, wbaer(naerb) = unset, gbaer(naerb) = unset
    real(kr) :: zbaer(65) = unset, dbaer(65) = unset
    real(kr) :: pmaer(naerb*299) = unset   ! IAER 5: phase-function moments 1.. per wavelength, wavelength index fastest
  end type

  ! aerosol.dat in full: "nn nmom", then per wavelength the wavelength and nn layer records (depth, albedo, moments)
  type aerosol_file
    integer :: nn = 0, nmom = 0
    integer :: nset = 0                                 ! wavelength headers in the file
    integer :: short = 0                                ! the set whose layer records ran out (0: none)
    real(kr), allocatable :: wl(:), tau(:, :), ssa(:, :), pm(:, :, :)    ! [set], [nn, set], [nmom, nn, set]
    ! per wavelength of the run (plan_aerosol_file): the two sets aeread would hold and the wavelengths it
    ! would carry for them: (1, :) the set read last, (2, :) the one before
    integer, allocatable :: pair(:, :)
    real(kr), allocatable :: pair_wl(:, :)
  end type

  type aerosol_load
    type(aerosol_file) :: file
    integer :: iaer = 0, imoma = 3, nosct = 0
    real(kr) :: abaer = 0
    integer :: nwl = 0                                  ! boundary-layer spectrum: wavelengths, extinction,
    real(kr), allocatable :: wl(:), ext(:), absb(:), asym(:)   ! absorption, asymmetry factor
    real(kr), allocatable :: column(:)                  ! per layer (1 = top): 0.55 um optical depth / ext(0.55)
    integer :: npm = 0                                  ! user phase-function moments per wavelength (then imoma = 0)
    real(kr), allocatable :: pm(:)                      ! [npm][nwl], wavelength index fastest
    integer :: nstrat = 0
    integer :: strat_layer(naerz) = 0, jaer(naerz) = 0
    real(kr) :: taerst(naerz) = 0
  end type

  real(kr), pointer, save :: t_awl(:) => null(), t_strat(:) => null()

contains

  integer function last_set(v) result(n)              ! largest index whose value is not "unset"
    real(kr), intent(in) :: v(:)
    integer :: i
    n = 0
    do i = 1, size(v)
      if (v(i) /= unset) n = i
    end do
  end function

  ! boundary-layer extinction, single-scattering albedo and asymmetry at wl: log-log interpolation of the
  ! spectrum, Angstrom law beyond its ends
  subroutine boundary_layer_at(a, wl, extinc, wa, ga)
    type(aerosol_load), intent(in) :: a
    real(kr), intent(in) :: wl
    real(kr), intent(out) :: extinc, wa, ga
    real(kr) :: wt, absorp
    integer :: l, n
    extinc = 0.; wa = 0.; ga = 0.
    if (a%iaer == 0) return
    n = a%nwl
    l = bracket(a%wl(1:n), wl)
    if (wl <= a%wl(1)) then
      extinc = a%ext(1)*(a%wl(1)/wl)**a%abaer
      wa = 1. - (a%absb(1)/a%ext(1))
      ga = a%asym(1)
    else if (wl >= a%wl(n)) then
      extinc = a%ext(n)*(a%wl(n)/wl)**a%abaer
      wa = 1. - (a%absb(n)/a%ext(n))
      ga = a%asym(n)
    else
      wt = log(wl/a%wl(l))/log(a%wl(l + 1)/a%wl(l))
      extinc = a%ext(l)*(a%ext(l + 1)/a%ext(l))**wt
      if (a%absb(l) > 0. .and. a%absb(l + 1) > 0.) then
        absorp = a%absb(l)*(a%absb(l + 1)/a%absb(l))**wt
      else
        absorp = a%absb(l)*(1. - wt) + a%absb(l + 1)*wt
      end if
      if (extinc > 0.) wa = max(0._kr, min(1._kr - absorp/extinc, 1._kr))
      ga = (1. - wt)*a%asym(l) + wt*a%asym(l + 1)
    end if
  end subroutine

  ! stratospheric model ja at wl: extinction relative to 0.55 um, single-scattering albedo, asymmetry
  subroutine stratospheric_at(abaer, ja, wl, qa, wa, ga)
    real(kr), intent(in) :: abaer, wl
    integer, intent(in) :: ja
    real(kr), intent(out) :: qa, wa, ga
    real(kr), pointer :: awl(:), t(:)
    real(kr) :: wt, absorp
    integer :: l
    awl => t_awl; t => t_strat
    wa = 0.
    l = bracket(awl, wl)
    if (wl <= awl(1)) then
      qa = s(1, 1)*(awl(1)/wl)**abaer
      wa = 1. - (s(1, 2)/s(1, 1))
      ga = s(l, 3)
    else if (wl >= awl(naerw)) then
      qa = s(naerw, 1)*(awl(1)/wl)**abaer               ! (the reference scales from the FIRST wavelength here)
      wa = 1. - (s(naerw, 2)/s(naerw, 1))
      ga = s(naerw, 3)
    else
      wt = log(wl/awl(l))/log(awl(l + 1)/awl(l))
      qa = s(l, 1)*(s(l + 1, 1)/s(l, 1))**wt
      absorp = s(l, 2)*(s(l + 1, 2)/s(l, 2))**wt
      if (qa > 0.) wa = max(0._kr, min(1._kr - absorp/qa, 1._kr))
      ga = (1. - wt)*s(l, 3) + wt*s(l + 1, 3)
    end if
  contains
    real(kr) function s(iw, q)
      integer, intent(in) :: iw, q
      s = t(iw + (q - 1)*naerw + (ja - 1)*3*naerw)
    end function
  end subroutine

  function new_aerosol_load(in, z, rh_surface) result(a)
    type(aerosol_input), intent(in) :: in
    real(kr), intent(in) :: z(:), rh_surface
    type(aerosol_load) :: a
    real(kr), parameter :: visfac = 3.912, floor = .00000001
    real(kr), allocatable :: zb(:), db(:)
    real(kr), pointer :: te(:), ta(:), tg(:), rhz(:)
    character(len=5) :: model
    real(kr) :: rhaer, rhum, wt, ext55, w55, g55, sigma, tbaer, zu, zd, f
    integer :: nz, nzb, ndb, i, j, lev, nq, nw, ng, ne
    character(len=80) :: errmes(10)
    nz = size(z)
    t_awl => tbl('aer.wl'); t_strat => tbl('aer.strat')       ! (resolved once per run)
    a%iaer = in%iaer; a%imoma = in%imoma; a%nosct = in%nosct; a%abaer = in%abaer
    a%jaer = in%jaer; a%taerst = in%taerst
    ! ---- stratospheric layers: the layer that holds each altitude ----
    if (any(in%jaer /= 0)) then
      do i = 1, naerz
        if (in%taerst(i) /= 0.) a%nstrat = i
      end do
      if (a%nstrat > 0) call layers_of_altitudes(z, in%zaer(1:a%nstrat), a%strat_layer(1:a%nstrat))
    end if
    if (in%iaer == 0) return
    if (in%iaer == -1) then
      a%file = read_aerosol_file(nz)
      return
    end if

    ! ---- vertical profile of the boundary-layer aerosol: user's or standard ----
    nzb = last_set(in%zbaer); ndb = last_set(in%dbaer)
    if (ndb > 0) then
      if (nzb == 1) then
        write(*, *) 'Error -- only one value of zbaer set'
        stop
      end if
      if (nzb /= ndb .and. nzb > 1) then
        write(*, *) 'Error -- number of elements must match:'
        write(*, '(a,/,(10es11.3))') 'zbaer', in%zbaer(1:nzb)
        write(*, '(a,/,(10es11.3))') 'dbaer', in%dbaer(1:ndb)
        stop
      end if
      db = in%dbaer(1:ndb)
      if (nzb == 0) then
        zb = z(1:ndb)
      else
        zb = in%zbaer(1:nzb)
      end if
    else
      zb = tbl('aer.z'); db = tbl('aer.density')
    end if

    ! ---- spectrum ----
    tbaer = in%tbaer
    if (in%iaer == 5) then
      call user_spectrum(ext55)
      if (in%vis == unset .and. tbaer == unset) tbaer = ext55
    else
      a%abaer = 0.
      rhaer = in%rhaer
      if (rhaer < 0.) rhaer = rh_surface
      rhum = max(0._kr, min(1._kr, rhaer))
      rhz => tbl('aer.rhzone')
      if (rhum < rhz(2)) then
        j = 1
      else if (rhum < rhz(3)) then
        j = 2
      else
        j = 3
      end if
      wt = (rhum - rhz(j))/(rhz(j + 1) - rhz(j))
      select case (abs(in%iaer))
      case (1); model = 'rural'
      case (2); model = 'urban'
      case (3); model = 'ocean'
      case default; model = 'tropo'
      end select
      te => tbl('aer.'//model//'.e'); ta => tbl('aer.'//model//'.a'); tg => tbl('aer.'//model//'.g')
      a%nwl = naerw
      allocate(a%wl(naerw), a%ext(naerw), a%absb(naerw), a%asym(naerw))
      a%wl = tbl('aer.wl')
      do i = 1, naerw
        a%ext(i) = between(te(i + (j - 1)*naerw), te(i + j*naerw))
        a%absb(i) = between(ta(i + (j - 1)*naerw), ta(i + j*naerw))
        a%asym(i) = between(tg(i + (j - 1)*naerw), tg(i + j*naerw))
      end do
      call boundary_layer_at(a, wl55, ext55, w55, g55)
      if (in%vis == unset .and. tbaer == unset) then
        print *, 'must specify either tbaer or vis'
        stop
      end if
    end if

    ! ---- amount per layer: density at the layer's lower level x thickness, above the top level a 5 km
    !      scale height; normalised to TBAER or to the surface extinction 3.912 / VIS ----
    allocate(a%column(nz))
    a%column(1) = density_at(100._kr)*5.
    zu = z(nz)
    do i = 2, nz
      lev = nz - i + 1
      zd = z(lev)
      a%column(i) = (zu - zd)*density_at(zd)
      zu = zd
    end do
    sigma = 0.
    if (ext55 > 0.) then
      if (tbaer >= 0) then
        if (sum(a%column) /= 0) sigma = tbaer/(ext55*sum(a%column))
      else
        sigma = visfac/(ext55*in%vis*density_at(0._kr))
      end if
    end if
    a%column = sigma*a%column
  contains
    real(kr) function between(v1, v2) result(v)          ! geometric interpolation in humidity, floored
      real(kr), intent(in) :: v1, v2
      real(kr) :: e1, e2
      e1 = max(v1, floor); e2 = max(v2, floor)
      v = e1*(e2/e1)**wt
    end function
    real(kr) function density_at(zz) result(d)
      real(kr), intent(in) :: zz
      real(kr) :: zc
      integer :: k
      zc = max(0._kr, min(100._kr, zz))
      d = 0.
      if (zc > zb(size(zb))) return
      k = bracket(zb, zc)
      f = (zc - zb(k))/(zb(k + 1) - zb(k))
      if (min(db(k), db(k + 1)) <= 0._kr) then
        d = max(db(k)*(1. - f) + db(k + 1)*f, 0.0_kr)
      else
        d = db(k)*(db(k + 1)/db(k))**f
      end if
    end function
    subroutine user_spectrum(q55)                        ! IAER=5 (usraer)
      real(kr), intent(out) :: q55
      real(kr) :: wlb(naerb), qb(naerb), wb(naerb), gb(naerb)
      real(kr), allocatable :: pmu(:)
      integer :: nwlb, k, npm
      wlb = in%wlbaer; qb = in%qbaer; wb = in%wbaer; gb = in%gbaer
      nwlb = last_set(wlb); nq = last_set(qb); nw = last_set(wb); ng = last_set(gb)
      npm = last_set(in%pmaer)
      ne = 0
      if (nwlb == 0) then
        qb(1) = 1.; nq = 1
        if (nw /= 1) call complain('specify one value of wbaer when wlbaer not set')
      else if (nwlb == 1) then
        if (nq > 1) then
          call complain('number of elements must match: wlbaer, qbaer')
        else if (nq == 0) then
          qb(1) = 1.
        end if
        if (nw /= 1) call complain('number of elements must match: wlbaer, wbaer')
      else
        if (nwlb /= nq) call complain('number of elements must match: wlbaer, qbaer')
        if (nwlb /= nw) call complain('number of elements must match: wlbaer, wbaer')
        if (ng == 0) then
          if (npm >= 1) then
            if (mod(npm, nwlb) /= 0) then
              call complain('incorrect number of phase function moments')
            else
              npm = npm/nwlb
            end if
          else
            call complain('must specify either gbaer or pmaer')
          end if
        else if (ng /= nw) then
          call complain('number of elements must match: wlbaer, gbaer')
        end if
      end if
      if (((ng == 0) .eqv. (npm == 0)) .and. in%imoma == 3) call complain('must specify either gbaer or pmaer, not both')
      if (ne > 0) then
        write(*, *) 'Error in user specified aerosols (iaer=5)'
        write(*, '(/,1x,5a8)') 'nwlbaer', 'nqbaer', 'nwbaer', 'ngbaer', 'npmaer'
        write(*, '(5i8,/)') nwlb, nq, nw, ng, npm
        do k = 1, ne
          write(*, '(2a)') 'Error in USRAER -- ', errmes(k)
        end do
        stop
      end if
      if (nw == 1) then                                  ! one point: a second one an octave up, Angstrom law
        if (nwlb == 0) wlb(1) = wl55
        wlb(2) = 2*wlb(1)
        nwlb = 2
        qb(2) = qb(1)*(wlb(1)/wlb(2))**in%abaer
        wb(2) = wb(1)
        gb(2) = gb(1)
        if (npm > 0) then                                ! the same moments at both points
          allocate(pmu(2*npm))
          pmu(1:2*npm - 1:2) = in%pmaer(1:npm)
          pmu(2:2*npm:2) = in%pmaer(1:npm)
        end if
      else if (npm > 0) then
        pmu = in%pmaer(1:npm*nwlb)
      end if
      if (npm > 0) then
        a%npm = npm
        a%imoma = 0
        call move_alloc(pmu, a%pm)
      end if
      a%nwl = nwlb
      allocate(a%wl(nwlb), a%ext(nwlb), a%absb(nwlb), a%asym(nwlb))
      a%wl = wlb(1:nwlb)
      a%ext = qb(1:nwlb)
      a%absb = (1. - wb(1:nwlb))*a%ext
      a%asym = gb(1:nwlb)
      k = bracket(a%wl, wl55)
      f = log(wl55/wlb(k))/log(wlb(k + 1)/wlb(k))
      if (wl55 < wlb(1)) then
        q55 = qb(1)*(wlb(1)/wl55)**in%abaer
      else if (wl55 > wlb(nwlb)) then
        q55 = qb(nwlb)*(wlb(nwlb)/wl55)**in%abaer
      else
        q55 = qb(k)*(qb(k + 1)/qb(k))**f
      end if
    end subroutine
    subroutine complain(msg)
      character(len=*), intent(in) :: msg
      ne = ne + 1
      errmes(ne) = msg
    end subroutine
  end function

  ! aerosol optical depth and single-scattering albedo of every layer (1 = top) at wl; the aerosols' part of
  ! the un-normalised phase-function moments (moment x scattering depth) is ADDED to pmom
  ! aerosol.dat, every record of it (aeread reads it as the wavelength loop advances: same list-directed reads,
  ! one record per "nn nmom", per wavelength and per layer)
  function read_aerosol_file(nz) result(f)
    integer, intent(in) :: nz
    type(aerosol_file) :: f
    real(kr) :: w
    real(kr), allocatable :: row(:)
    integer :: u, ios, i, pass, n
    open(newunit=u, file='aerosol.dat', form='formatted', status='old', iostat=ios)
    if (ios /= 0) then
      write(*, *) 'iaer=-1: aerosol.dat not found in the run directory'
      stop 1
    end if
    read(u, *, iostat=ios) f%nn, f%nmom
    if (ios /= 0) then
      write(*, *) 'no data found in aerosol.dat'
      stop 1
    end if
    if (nz - f%nn + 1 <= 0) then
      write(*, *) 'nz  nn ', nz, f%nn
      write(*, *) 'too many layers specified in aerosol.dat'
      stop 1
    end if
    allocate(row(2 + f%nmom))
    do pass = 1, 2                                       ! count the sets, then keep them
      n = 0
      sets: do
        read(u, *, iostat=ios) w
        if (ios /= 0) exit
        n = n + 1
        if (pass == 2) f%wl(n) = w
        do i = 1, f%nn
          read(u, *, iostat=ios) row
          if (ios /= 0) then
            f%short = n
            exit sets
          end if
          if (pass == 2) then
            f%tau(i, n) = row(1); f%ssa(i, n) = row(2); f%pm(:, i, n) = row(3:)
          end if
        end do
      end do sets
      if (pass == 1) then
        f%nset = n
        allocate(f%wl(n), f%tau(f%nn, n), f%ssa(f%nn, n), f%pm(f%nmom, f%nn, n))
        f%tau = 0.; f%ssa = 0.; f%pm = 0.
        rewind u
        read(u, *)
      end if
    end do
    close(u)
    if (f%nset == 0) then
      write(*, *) 'no data found in aerosol.dat'
      stop 1
    end if
  end function

  ! Which two sets of aerosol.dat serve each wavelength of the run.  aeread (tauaero.f:1607-1672) keeps two
  ! slots and reads forward while the wavelength is beyond both; what it holds at a call depends on the calls
  ! before it (a first wavelength at or below the file's first makes the first set spectrally uniform, carried
  ! at half and twice THAT wavelength until the run passes twice it), so the wavelengths are walked in order,
  ! once, ahead of the parallel loop.
  subroutine plan_aerosol_file(a, wls)
    type(aerosol_load), intent(inout) :: a
    real(kr), intent(in) :: wls(:)
    real(kr) :: w(2), wl0, wl
    integer :: d(2), ind, more, next, iw
    logical :: at_end
    if (a%iaer /= -1) return
    associate (f => a%file)
      allocate(f%pair(2, size(wls)), f%pair_wl(2, size(wls)))
      w = 0.; d = 0; ind = 1; more = 1; wl0 = 0.; next = 1
      do iw = 1, size(wls)
        wl = wls(iw)
        if (w(1) == 0.) then
          call take(1, 1)
          wl0 = w(1); ind = 1; next = 2
        else if (wl < minval(w) .and. wl > wl0) then     ! back to the first set
          call take(1, 1)
          w(2) = 0.; ind = 1; more = 1; next = 2
        end if
        if (more == 1) then
          at_end = .false.
          do while (wl > maxval(w))
            more = 0
            if (next > f%nset) then
              at_end = .true.
              exit
            end if
            ind = 3 - ind
            call take(ind, next)
            next = next + 1
          end do
          if (.not. at_end) more = 1
        end if
        if (w(2) == 0.) then                             ! one set so far: spectrally uniform
          w(1) = .5*wl; w(2) = 2*wl
          d(2) = d(1)
          wl0 = 0.
        end if
        f%pair(:, iw) = [d(ind), d(3 - ind)]
        f%pair_wl(:, iw) = [w(ind), w(3 - ind)]
      end do
    end associate
  contains
    subroutine take(slot, set)
      integer, intent(in) :: slot, set
      if (a%file%short == set) then
        write(*, *) 'not enough aerosol records'
        write(*, *) 'need ', a%file%nn, ' records'
        stop 1
      end if
      d(slot) = set; w(slot) = a%file%wl(set)
    end subroutine
  end subroutine

  ! The run's aerosols as scattering terms of the compact batch form (include/sbdart_amd.h, sbd_mix_in): slot 1 the
  ! boundary layer's (family = IMOMA) when there is one, then one slot per active stratospheric layer (Henyey-
  ! Greenstein).  ok = .false.: moments that are not a function of one asymmetry factor (aerosol.dat, the user's own
  ! moments) -- such runs keep the arrays form.
  subroutine aerosol_terms(a, nterm, family, ok)
    type(aerosol_load), intent(in) :: a
    integer, intent(out) :: nterm, family(:)
    logical, intent(out) :: ok
    integer :: i
    nterm = 0; family = 0
    ok = a%iaer /= -1
    if (a%iaer /= 0 .and. a%iaer /= -1) then
      ok = a%imoma >= 1 .and. a%imoma <= 3
      nterm = 1
      family(1) = a%imoma
    end if
    do i = 1, a%nstrat
      if (a%jaer(i) /= 0 .and. a%taerst(i) > 0.) then
        nterm = nterm + 1
        if (nterm > size(family)) then
          ok = .false.
          return
        end if
        family(nterm) = 3
      end if
    end do
  end subroutine

  ! trm (optional, slots as aerosol_terms counts them): per layer and slot the asymmetry factor and the two factors
  ! the term's moments are multiplied with, in the reference's order: PM*DTAUA*WAER (tauaero.f:1300), PM*DT*WA (1330)
  subroutine aerosol_depths(a, wl, nz, nmom, dtaua, waer, pmom, iw, trm)
    type(aerosol_load), intent(in) :: a
    real(kr), intent(in) :: wl
    integer, intent(in) :: nz, nmom
    real(kr), intent(out) :: dtaua(nz), waer(nz)
    real(kr), intent(inout) :: pmom(0:nmom, nz)
    integer, intent(in), optional :: iw                  ! index of wl among the run's wavelengths (aerosol.dat)
    real(kr), intent(out), optional :: trm(:, :, :)      ! (nz, 3, slots)
    real(kr) :: pm(0:nmom), extinc, wa, ga, dt, wt, ta, tb, gg
    integer :: i, j, nl, namom, l, ia, ib, k, slot
    dtaua = 0.; waer = 0.
    slot = 0
    if (present(trm)) trm = 0.
    if (a%iaer == -1) then
      ! layers nz-nn+1 .. nz from the file (the layers above them: nothing; the reference leaves them unset);
      ! depth log-log between the two sets where both are positive, everything else linear in the weight
      associate (f => a%file)
        ia = f%pair(1, iw); ib = f%pair(2, iw)
        wt = log(wl/f%pair_wl(1, iw))/log(f%pair_wl(2, iw)/f%pair_wl(1, iw))
        wt = max(0._kr, min(wt, 1._kr))
        do k = 1, f%nn
          i = nz - f%nn + k
          ta = f%tau(k, ia); tb = f%tau(k, ib)
          if (min(ta, tb) > 0.) then
            dtaua(i) = ta*(tb/ta)**wt
          else
            dtaua(i) = ta*(1. - wt) + tb*wt
          end if
          waer(i) = f%ssa(k, ia)*(1. - wt) + f%ssa(k, ib)*wt
          if (f%nmom == 1) then
            namom = nmom
            gg = f%pm(1, k, ia)*(1. - wt) + f%pm(1, k, ib)*wt
            call phase_moments(a%imoma, gg, nmom, pm)
          else
            namom = min(f%nmom, nmom)
            pm(1:namom) = f%pm(1:namom, k, ia)*(1. - wt) + f%pm(1:namom, k, ib)*wt
          end if
          pmom(1:namom, i) = pmom(1:namom, i) + pm(1:namom)*dtaua(i)*waer(i)
        end do
      end associate
    else if (a%iaer /= 0) then
      call boundary_layer_at(a, wl, extinc, wa, ga)
      if (a%nosct == 1) extinc = extinc*(1. - wa)
      if (a%nosct == 3) extinc = extinc*(1. - wa*ga)
      if (a%nosct /= 0) then
        wa = 0.; ga = 0.
      end if
      if (a%imoma > 0) then
        namom = nmom
        call phase_moments(a%imoma, ga, nmom, pm)
      else                                               ! the user's moments, linear in wavelength, end values outside
        namom = min(a%npm, nmom)
        l = bracket(a%wl(1:a%nwl), wl)
        wt = (wl - a%wl(l))/(a%wl(l + 1) - a%wl(l))
        wt = max(0._kr, min(wt, 1._kr))
        do j = 1, namom
          pm(j) = a%pm(l + (j - 1)*a%nwl)*(1. - wt) + a%pm(l + 1 + (j - 1)*a%nwl)*wt
        end do
      end if
      do i = 1, nz
        dtaua(i) = extinc*a%column(i)
        waer(i) = wa
        do j = 1, namom
          pmom(j, i) = pmom(j, i) + pm(j)*dtaua(i)*waer(i)
        end do
      end do
      slot = 1
      if (present(trm)) then
        trm(:, 1, 1) = ga; trm(:, 2, 1) = dtaua; trm(:, 3, 1) = waer
      end if
    end if
    do i = 1, a%nstrat
      if (a%jaer(i) /= 0 .and. a%taerst(i) > 0.) then
        nl = a%strat_layer(i)
        call stratospheric_at(a%abaer, a%jaer(i), wl, extinc, wa, ga)
        dt = a%taerst(i)*extinc
        call phase_moments(3, ga, nmom, pm)
        do j = 1, nmom
          pmom(j, nl) = pmom(j, nl) + pm(j)*dt*wa
        end do
        slot = slot + 1
        if (present(trm)) then
          trm(nl, 1, slot) = ga; trm(nl, 2, slot) = dt; trm(nl, 3, slot) = wa
        end if
        waer(nl) = (waer(nl)*dtaua(nl) + wa*dt)/(dtaua(nl) + dt)
        dtaua(nl) = dtaua(nl) + dt
      end if
    end do
  end subroutine

end module sbd_aerosol_mod
