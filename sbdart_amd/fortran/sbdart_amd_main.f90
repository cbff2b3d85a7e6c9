! program sbdart_amd -- Fortran-2003 host of the MI355X engine.
!
! Drop-in shape of the reference executable (drt.f:90-563): reads ./INPUT (namelist
! &INPUT [&DINPUT], same variable names), writes the IOUT-specific text to stdout
! (drt.f:892-1165), writes SBDART_WARNING.NN files (disutil.f:278-325).  The wavelength
! loop itself (drt.f:425-561) is restructured for the GPU: all (wavelength, k-term) work
! items are assembled first, solved in ONE batched call through the C ABI
! (sbd_engine_solve_host -> HIP kernels), and stdout1's accumulation then walks the
! results in the reference's order.
!
! Scope note (SURVEY.md 8f N1): the band model that turns &INPUT into per-wavelength
! optical depths (taugas/tauaero/taucloud/spectra/atms) is not part of this round; the
! per-work-item optical properties are read from an "SBDREC1" optics file
! (environment SBD_OPTICS, default ./OPTICS.sbdrec) with exactly the DISORT arguments
! drt.f:541-546 passes.  Everything downstream of that -- engine, retry of NSTR,
! accumulation, output formats -- is this program.
program sbdart_amd
  use iso_c_binding
  use sbd_engine_mod
  use sbd_host_mod
  implicit none
  integer, parameter :: ncldz = 5, naerz = 5, naerb = 150, maxmom = 299, ndb = 20
  ! ---- &INPUT / &DINPUT (drt.f:200-215), same names, same defaults where they matter ----
  integer :: idatm = 4, isat = 0, nf = 2, iday = 0, isalb = 0, krhclr = 0, jaer(naerz) = 0, iaer = 0, &
             nothrm = -1, nosct = 0, kdist = 3, ngrid = 0, idb(ndb) = 0, iout = 10, nstr = 0, nzen = 0, &
             nphi = 0, imomc = 3, imoma = 3, ibcnd = 0, ipth = 0
  real(kr) :: amix = zip, wlinf = real(.55, kr), wlsup = real(.55, kr), wlinc = 0, sza = 0, csza = zip, solfac = 1, &
       time = 16, alat = real(-64.767, kr), alon = real(-64.067, kr), zpres = zip, pbar = zip, sclh2o = zip, &
       uw = zip, uo3 = zip, o3trp = zip, ztrp = 0, xrsc = 1, xn2 = zip, xo2 = zip, xco2 = zip, xch4 = zip, &
       xn2o = zip, xco = zip, xno2 = zip, xso2 = zip, xnh3 = zip, xno = zip, xhno3 = zip, xo4 = 1, &
       albcon = 0, sc(5) = huge(0.), zcloud(ncldz) = 0, tcloud(ncldz) = 0, lwp(ncldz) = 0, nre(ncldz) = 8, &
       rhcld = zip, zaer(naerz) = 0, taerst(naerz) = 0, vis = zip, rhaer = zip, tbaer = zip, &
       wlbaer(naerb) = zip, qbaer(naerb) = zip, abaer = 0, wbaer(naerb) = zip, gbaer(naerb) = zip, &
       pmaer(naerb*maxmom) = zip, zbaer(mxly) = zip, dbaer(mxly) = zip, zgrid1 = 1, zgrid2 = 30, &
       zout(2) = (/0._kr, 100._kr/), temis = 0, uzen(nstrms) = zip, vzen(nstrms) = 90, phi(nstrms) = zip, &
       saza = 180, ttemp = zip, btemp = zip, phi0 = 0, fisot = 0
  logical :: prnt(7) = .false., corint = .false., spowder = .false.
  namelist /input/ idatm, amix, isat, wlinf, wlsup, wlinc, sza, csza, solfac, nf, iday, time, alat, alon, &
       zpres, pbar, sclh2o, uw, uo3, o3trp, ztrp, xrsc, xn2, xo2, xco2, xch4, xn2o, xco, xno2, xso2, xnh3, &
       xno, xhno3, xo4, isalb, albcon, sc, zcloud, tcloud, lwp, nre, rhcld, krhclr, jaer, zaer, taerst, iaer, &
       vis, rhaer, tbaer, wlbaer, qbaer, abaer, wbaer, gbaer, pmaer, zbaer, dbaer, nothrm, nosct, kdist, &
       zgrid1, zgrid2, ngrid, idb, zout, iout, prnt, temis, nstr, nzen, uzen, vzen, nphi, phi, saza, imomc, &
       imoma, ttemp, btemp, corint, spowder
  namelist /dinput/ ibcnd, phi0, prnt, ipth, fisot, temis, nstr, nzen, uzen, vzen, nphi, phi, ttemp, btemp

  type(optics_t), allocatable :: recs(:)
  integer :: nrec, ios, i, j, k, nz, nwl, il, nstrsv, nmom, numu, ntry, lev_top, lev_bot, nlev, u11
  logical :: radcalc, onlyfl, all_levels
  character(len=1024) :: optics_path
  integer :: plen, pstat
  real(kr) :: wlinc_eff, wl, wvlo, wvhi, dwl
  real(kr), allocatable, target :: dtauc(:,:), ssalb(:,:), pmom(:,:,:), wvnmlo(:), wvnmhi(:), fbeam(:), &
       albedo(:), flux(:,:,:), uu(:,:,:,:), temper(:), umu(:), phiv(:)
  integer(c_int8_t), allocatable, target :: plank(:)
  integer(c_int32_t), allocatable, target :: status(:), level_out(:)
  real(kr), allocatable :: zlev(:), uur(:,:,:), rfldir(:), rfldn(:), flup(:)
  type(sbd_run_cfg) :: cfg
  type(sbd_batch_in) :: bin
  type(sbd_batch_out) :: bout
  type(c_ptr) :: eng
  integer(c_int) :: rc
  integer :: stall

  ! ---- read ./INPUT exactly like drt.f:220-231 ----
  open(newunit=u11, file='INPUT', status='old', iostat=ios)
  if (ios == 0) then
    read(u11, input, iostat=ios)
    if (ios /= 0) stop 'error: namelist block $INPUT not found'
    read(u11, dinput, iostat=ios)
    close(u11)
  else
    write(*, input)
    stop
  end if

  radcalc = any(iout == (/5, 6, 20, 21, 22, 23/))      ! drt.f:237-247
  onlyfl = .not. radcalc
  if (nstr == 0) then
    if (radcalc) then
      nstr = min(20, nstrms)
    else
      nstr = 4
    end if
  end if
  if (radcalc) call view_angles(nphi, phi, nzen, uzen, vzen, iout, nstr)
  if (iout == 22) call fatal('iout=22 (radiance at every level) is not wired in this host yet')
  phi0 = mod(saza - 180.0_kr + 360.0_kr, 360.0_kr)      ! drt.f:283

  ! ---- per-work-item optical properties (stand-in for gasset/taucloud/tauaero/rayleigh) ----
  call get_environment_variable('SBD_OPTICS', optics_path, plen, pstat)
  if (pstat /= 0 .or. plen <= 0) optics_path = 'OPTICS.sbdrec'
  call read_optics(trim(optics_path), recs, nrec)
  if (nrec < 1) call fatal('optics file holds no work items')
  nz = recs(1)%nlyr
  nmom = recs(1)%nmom
  nstrsv = nstr

  ! spectral grid size printed by stdout0 (setfilt, spectra.f:3370-3384; isat=0: wlmin=wlinf)
  wlinc_eff = wlinc
  nwl = grid_size(wlinf, wlsup, wlinc_eff)
  ! cross-check the band edges of the optics against wllimits (drt.f:1657-1740)
  do i = 1, nrec
    il = recs(i)%iwl - 1
    if (il < 0 .or. il >= nwl) call fatal('optics record outside the spectral grid of INPUT')
    call wl_limits(il, nwl, wlinc_eff, wlinf, wlsup, wl, wvlo, wvhi)
    if (abs(wl - recs(i)%wl) > 1e-12_kr*wl .or. abs(wvlo - recs(i)%wvnmlo) > 1e-9_kr*wvlo .or. &
        abs(wvhi - recs(i)%wvnmhi) > 1e-9_kr*wvhi) call fatal('optics record disagrees with the wavelength grid of INPUT')
  end do

  ! ---- batch arrays (row-major by work item == Fortran's first index fastest) ----
  allocate(dtauc(nz, nrec), ssalb(nz, nrec), pmom(0:nmom, nz, nrec), wvnmlo(nrec), wvnmhi(nrec), &
           fbeam(nrec), albedo(nrec), plank(nrec), status(nrec), temper(0:nz))
  do i = 1, nrec
    if (recs(i)%nlyr /= nz .or. recs(i)%nmom /= nmom) call fatal('optics records differ in NLYR/NMOM')
    dtauc(:, i) = recs(i)%dtauc
    ssalb(:, i) = recs(i)%ssalb
    pmom(:, :, i) = recs(i)%pmom
    wvnmlo(i) = recs(i)%wvnmlo; wvnmhi(i) = recs(i)%wvnmhi
    fbeam(i) = recs(i)%fbeam; albedo(i) = recs(i)%albedo
    plank(i) = int(iand(recs(i)%flags, 1), c_int8_t)
  end do
  temper = recs(1)%temper
  if (btemp < 0._kr) btemp = recs(1)%btemp          ! drt.f:334-335 defaults come with the profile
  if (ttemp < 0._kr) ttemp = recs(1)%ttemp

  ! output levels: ntop = 1 (TOA), nbot = nz+1 (surface) for zout = 0,100 (drt.f:376-381)
  all_levels = (iout == 7 .or. iout == 11)
  if (all_levels) then
    nlev = nz + 1
    lev_top = 1; lev_bot = nz + 1
    allocate(level_out(nlev))
    level_out = (/(i - 1, i = 1, nlev)/)
  else
    nlev = 2
    lev_top = 1; lev_bot = 2
    allocate(level_out(2))
    level_out = (/0, nz/)
  end if

  numu = 0
  if (radcalc) then                                   ! drt.f:391-403
    numu = nzen
    allocate(umu(numu), phiv(nphi))
    do j = 1, numu
      umu(j) = min(1._kr, max(cos(uzen(numu + 1 - j)*(real(3.1415926536d0, kr)/180._kr)), -1._kr))
      if (umu(j) == 0._kr) then
        if (j == numu) then
          umu(j) = -real(.0001, kr)
        else
          umu(j) = real(.0001, kr)
        end if
      end if
    end do
    phiv = phi(1:nphi)
    allocate(uurs(max(nzen, 1), max(nphi, 1)))
    uurs = 0
  else
    allocate(umu(1), phiv(1))
  end if
  allocate(fxdn(nz), fxup(nz), fxdir(nz))
  fxdn = 0; fxup = 0; fxdir = 0

  ! ---- engine, with the reference's NSTR "dithering" retry (drt.f:536-555) ----
  eng = c_null_ptr
  do ntry = 0, 2
    nstr = nstrsv + ntry*(3*ntry - 5)
    if (nstr < 4) cycle
    if (nstr > nstrms) exit
    cfg%abi_version = SBD_ABI_VER
    cfg%nlyr = nz; cfg%nstr = nstr; cfg%nmom = nmom
    cfg%onlyfl = merge(1, 0, onlyfl); cfg%lamber = 1; cfg%usrang = merge(1, 0, radcalc)
    cfg%numu = numu; cfg%nphi = merge(nphi, 0, radcalc)
    cfg%nlevel_out = nlev; cfg%device = 0; cfg%max_batch = nrec
    cfg%umu0 = recs(1)%umu0; cfg%phi0 = phi0; cfg%fisot = fisot
    cfg%btemp = btemp; cfg%ttemp = ttemp; cfg%temis = temis
    cfg%temper = c_loc(temper); cfg%umu = c_loc(umu); cfg%phi = c_loc(phiv)
    cfg%level_out = c_loc(level_out)
    rc = sbd_engine_create(cfg, eng)
    if (rc == SBD_OK) exit
    if (rc == SBD_E_RETRY_NSTR) then
      call warn_file(1, 'SETDIS--beam angle=computational angle; change NSTR')
      if (any(fbeam > 0._kr)) then
        call sbd_engine_destroy(eng)
        eng = c_null_ptr
        cycle
      end if
      exit
    end if
    call fatal('sbd_engine_create: '//sbd_strerror_f(rc)//' '//sbd_last_error_f())
  end do
  if (.not. c_associated(eng)) then
    write(*, *) 'Error --- NSTR dithering procedure failed'
    stop
  end if

  allocate(flux(nlev, SBD_NFLUX, nrec))
  if (radcalc) then
    allocate(uu(numu, nlev, nphi, nrec))
  else
    allocate(uu(1, 1, 1, 1))
  end if
  bin%nwork = nrec
  bin%dtauc = c_loc(dtauc); bin%ssalb = c_loc(ssalb); bin%pmom = c_loc(pmom)
  bin%wvnmlo = c_loc(wvnmlo); bin%wvnmhi = c_loc(wvnmhi); bin%fbeam = c_loc(fbeam)
  bin%albedo = c_loc(albedo); bin%plank = c_loc(plank)
  bout%flux = c_loc(flux); bout%status = c_loc(status)
  bout%uu = c_null_ptr
  if (radcalc) bout%uu = c_loc(uu)

  ! the wavelength loop, one batched call (filter ff = 0 items are solved too; their weight is 0)
  rc = sbd_engine_solve_host(eng, bin, bout)
  if (rc /= SBD_OK) call fatal('sbd_engine_solve_host: '//sbd_strerror_f(rc)//' '//sbd_last_error_f())

  ! ---- warnings / fatals the reference raises through errmsg ----
  stall = 0
  do i = 1, nrec
    stall = ior(stall, status(i))
  end do
  if (iand(stall, SBD_ST_ERR_INPUT) /= 0) call warn_file(0, 'DISORT--input and/or dimension errors')
  if (iand(stall, SBD_ST_ERR_EIGEN) /= 0) call warn_file(0, 'ASYMTX--convergence problems')
  if (iand(stall, SBD_ST_WARN_SOLVE0) /= 0) call warn_file(2, 'SOLVE0--SGBCO says matrix near singular')
  if (iand(stall, SBD_ST_WARN_UPBEAM) /= 0) call warn_file(3, 'UPBEAM--SGECO says matrix near singular')
  if (iand(stall, SBD_ST_WARN_UPISOT) /= 0) call warn_file(4, 'UPISOT--SGECO says matrix near singular')
  if (iand(stall, SBD_ST_WARN_PLKAVG) /= 0) call warn_file(10, 'PLKAVG--returns zero; possible underflow')
  if (any(plank /= 0)) then                         ! CHEKIN warning 6 (disort.f:5145-5152)
    do i = 1, nz
      if (abs(temper(i) - temper(i - 1)) > 10._kr) then
        call warn_file(6, 'CHEKIN--vertical temperature step may be too large for good accuracy')
        exit
      end if
    end do
  end if
  if (radcalc .and. any(fbeam > 0._kr)) &          ! CHEKIN warning 7 (disort.f:5154-5158)
    call warn_file(7, 'CHEKIN--intensity correction is off; intensities may be less accurate')

  ! ---- stdout0 / stdout1 / stdout2 in the reference's order ----
  allocate(zlev(nz), rfldir(nlev), rfldn(nlev), flup(nlev))
  zlev = 0
  if (radcalc) then
    allocate(uur(numu, nlev, nphi))
  else
    allocate(uur(1, 1, 1))
  end if
  call stdout0(iout, nwl, nz)
  do i = 1, nrec
    dwl = 10000._kr/wvnmlo(i) - 10000._kr/wvnmhi(i)      ! drt.f:438
    rfldir = flux(:, 1, i); rfldn = flux(:, 2, i); flup = flux(:, 3, i)
    if (radcalc) uur = uu(:, :, :, i)
    call stdout1(nz, zlev, lev_top, lev_bot, iout, recs(i)%wl, dwl, recs(i)%wt, rfldir, rfldn, flup, &
                 recs(i)%ff, nphi, nzen, phi, uzen, uur, lev_top, lev_bot, recs(i)%kd, recs(i)%nk)
  end do
  call stdout2(iout, wlinf, wlsup, nphi, nzen, phi, uzen)
  call sbd_engine_destroy(eng)

contains
  subroutine fatal(msg)
    character(len=*), intent(in) :: msg
    write(0, '(a)') 'sbdart_amd: '//msg
    stop 1
  end subroutine
end program sbdart_amd
