! program sbdart_amd -- Fortran-2003 host of the MI355X engine.
!
! Drop-in shape of the reference executable (drt.f:90-563): reads ./INPUT (namelist
! &INPUT [&DINPUT], same variable names), writes the IOUT-specific text to stdout
! (drt.f:892-1165), writes SBDART_WARNING.NN files (disutil.f:278-325).  The wavelength
! loop itself (drt.f:425-561) is restructured for the GPU: all (wavelength, k-term) work
! items are assembled first and solved in ONE batched call through the C ABI
! (sbd_fleet_solve_host: the batch is sharded over every visible GPU, HIP kernels per shard);
! the per-run output formats take stdout1's weighted sums from the engine's reduction (one
! RCCL reduce between GPUs), the per-wavelength formats sum the <= 3 k-terms of a point here.
!
! The per-work-item optical properties (exactly the DISORT arguments drt.f:541-546 passes) come from
! the band model of this host (sbd_bandmodel_mod and the modules it uses: atmospheres, LOWTRAN7 gases with
! the 3-term k-distribution, Rayleigh, clouds, aerosols, surface and solar spectra, sensor filters --
! SURVEY.md 8f N1) or, for the few inputs it does not cover (it says which), from an "SBDREC1" optics
! file the reference produced (environment SBD_OPTICS, default ./OPTICS.sbdrec; an optics file, when
! present, always wins).  Everything downstream of that
! -- engine, retry of NSTR, accumulation, output formats -- is this program.
!
! Round 4: the run is a procedure (run_once) so that ONE process can serve many runs -- `sbdart_amd --batch LIST`
! (LIST: one run directory per line, each holding its INPUT) for the harnesses that launch hundreds of tiny runs
! (RunRT's sweeps, RunRT/RunRT.py:2158-2193; TestRuns/test_runs): the HIP runtime comes up once, the fleets are kept
! and found again by configuration, every run's text goes to SBDART.stdout in its directory.  A run of a batch is made
! in two phases: phase 1 (INPUT -> screening -> band model -> work items) in a forked child per run, where the
! reference's STOP semantics cost nothing and a pool of workers runs them side by side; phase 2 (work items -> engine
! -> output records) in the one process that owns the GPU.
module sbd_fleet_cache_mod
  use iso_c_binding
  use sbd_grid_mod, only: kr
  implicit none
  integer, parameter :: max_fleets = 48
  type fleet_slot
    type(c_ptr) :: fleet = c_null_ptr
    integer(c_int) :: rc = 0
    integer :: max_batch = 0, age = 0
    real(kr), allocatable :: key(:)
  end type
  type(fleet_slot), save :: slots(max_fleets)
  integer, save :: nslot = 0, clock = 0
  integer(c_int32_t), allocatable, target, save :: devices(:)
end module

module sbd_run_mod
  implicit none
  real(kind=8), save :: t_engine = 0, t_wait = 0, t_phase2 = 0        ! batch mode's time account (SBD_TIMING)
  integer(kind=8), save :: tick_program = -1                          ! system_clock at the program's first statement
  real(kind=8), save :: t_create = 0                                  ! sbd_fleet_create calls of the run (SBD_TIMING)
  ! sbdart_amd --serve: serve_mode = this process (or the run's forked first phase) works for a client of the resident
  ! server -- messages go to file descriptor 2, which is the CLIENT's; serve_big = the estimate of the run's work items (bytes)
  ! above which the forked first phase hands the run back to be made whole in the server (exit code 77)
  integer, parameter :: gas_device_min = 8000       ! spectral points from which a run's gas terms are evaluated on the device
  logical, save :: serve_mode = .false.
  real(kind=8), save :: serve_big = 64.0d6, serve_mid = 1.0d6
  character(len=*), parameter :: items_tmp = '.sbd_items.part', items_file = '.sbd_items', stdout_file = 'SBDART.stdout', stderr_file = 'SBDART.stderr', phase1_mark = '.sbd_phase1'
contains

! phase 0: the whole run, as the reference's executable; 1: up to the work items (written to items_file); 2: from there;
! 3: the whole run inside a process that serves further runs (sbdart_amd --serve, runs too large for a file of work
!    items): like 0 -- compact form, gas terms on the device -- but an error ends the run, not the process, like 2
subroutine run_once(phase)
  use iso_c_binding
  use sbd_engine_mod
  use sbd_grid_mod
  use sbd_io_mod
  use sbd_output_mod
  use sbd_atmos_mod, only: atmosphere
  use sbd_bandmodel_mod
  use sbd_ckfile_mod, only: ck_file, read_ck_files
  use sbd_tables_mod, only: tables_load, tables_image
  use sbd_filter_mod
  use sbd_fleet_cache_mod
  use omp_lib, only: omp_get_max_threads
  integer, intent(in) :: phase
  interface
    subroutine sbd_px_exit_now(code) bind(C, name='sbd_px_exit_now')
      import; integer(c_int), value :: code
    end subroutine
  end interface
  integer, parameter :: ncldz = 5, naerz = 5, naerb = 150, maxmom = 299, ndb = 20
  ! ---- &INPUT / &DINPUT (drt.f:200-215), same names, same defaults where they matter ----
  integer :: idatm, isat, nf, iday, isalb, krhclr, jaer(naerz), iaer, nothrm, nosct, kdist, ngrid, idb(ndb), iout, nstr, &
             nzen, nphi, imomc, imoma, ibcnd, ipth
  real(kr) :: amix, wlinf, wlsup, wlinc, sza, csza, solfac, time, alat, alon, zpres, pbar, sclh2o, uw, uo3, o3trp, ztrp, &
       xrsc, xn2, xo2, xco2, xch4, xn2o, xco, xno2, xso2, xnh3, xno, xhno3, xo4, albcon, sc(5), zcloud(ncldz), &
       tcloud(ncldz), lwp(ncldz), nre(ncldz), rhcld, zaer(naerz), taerst(naerz), vis, rhaer, tbaer, wlbaer(naerb), &
       qbaer(naerb), abaer, wbaer(naerb), gbaer(naerb), pmaer(naerb*maxmom), zbaer(mxly), dbaer(mxly), zgrid1, zgrid2, &
       zout(2), temis, uzen(nstrms), vzen(nstrms), phi(nstrms), saza, ttemp, btemp, phi0, fisot
  logical :: prnt(7), corint, spowder
  namelist /input/ idatm, amix, isat, wlinf, wlsup, wlinc, sza, csza, solfac, nf, iday, time, alat, alon, &
       zpres, pbar, sclh2o, uw, uo3, o3trp, ztrp, xrsc, xn2, xo2, xco2, xch4, xn2o, xco, xno2, xso2, xnh3, &
       xno, xhno3, xo4, isalb, albcon, sc, zcloud, tcloud, lwp, nre, rhcld, krhclr, jaer, zaer, taerst, iaer, &
       vis, rhaer, tbaer, wlbaer, qbaer, abaer, wbaer, gbaer, pmaer, zbaer, dbaer, nothrm, nosct, kdist, &
       zgrid1, zgrid2, ngrid, idb, zout, iout, prnt, temis, nstr, nzen, uzen, vzen, nphi, phi, saza, imomc, &
       imoma, ttemp, btemp, corint, spowder
  namelist /dinput/ ibcnd, phi0, prnt, ipth, fisot, temis, nstr, nzen, uzen, vzen, nphi, phi, ttemp, btemp

  type(optics_t), allocatable :: recs(:)
  type(spectral_grid) :: grid
  type(view_geometry) :: view
  type(iout_format) :: fmt
  type(spectral_sums) :: sums
  integer :: nrec, ios, i, j, nz, nmom, numu, lev_top, lev_bot, nlev, u11, ntop, nbot, i0, i1, nbeam, npart, ip, pass, ncorr
  logical :: radcalc, known, have_atm, default_zout
  character(len=1024) :: path
  integer :: plen, pstat
  real(kr) :: wl, wvlo, wvhi, dwl
  real(kr), allocatable, target :: dtauc(:,:), ssalb(:,:), pmom(:,:,:), wvnmlo(:), wvnmhi(:), fbeam(:), &
       albedo(:), flux(:,:,:), uu(:,:,:,:), temper(:), umu(:), phiv(:), weight(:), acc_flux(:,:), acc_uu(:,:,:)
  integer(c_int8_t), allocatable, target :: plank(:)
  real(kr), allocatable, target :: bitem(:, :)        ! ocean surface: nr, ni, rsw per work item
  integer(c_int32_t), allocatable, target :: pmom_row(:)   ! block of moments per work item (band model: per wavelength)
  real(kr), allocatable, target :: pt_lo(:), pt_hi(:), pt_fb(:), pt_al(:)   ! compact form: per SPECTRAL POINT
  integer(c_int8_t), allocatable, target :: pt_pl(:)
  integer(c_int32_t), allocatable, target :: status(:), level_out(:)
  integer, allocatable :: order(:), where_solved(:), part_of(:)
  real(kr), allocatable :: zlev(:), plev(:)
  integer :: stall, fatal_at, nbad
  type(model_input) :: model
  type(sensor_filter) :: sensor
  type(atmosphere) :: atm
  logical :: have_file, ok, from_model, in_place, sum_widths, aborted
  real(kr), allocatable, target :: bdtauc(:, :), bssalb(:, :), bpmom(:, :, :)
  real(kr), allocatable :: btemper(:)
  integer(kind=8) :: tick0, tick1, tick2, tick_rate, tick_bm0, tick_bm1, tick_out, tick_c0, tick_c1
  character(len=256) :: why
  type(ck_file) :: ck                                  ! KDIST = -1: the k-distribution file pair
  type(mix_batch), target :: mix                       ! the run's batch in compact form (sbd_mix_in), when it fits
  logical :: use_mix, gas_dev, scat_dev, items_wanted
  real(kr), allocatable :: one_dtau(:), one_ssalb(:), one_pmom(:, :)
  ! the gas terms on the device (sbd_fleet_gas_terms): per spectral point the number of k-terms, their weights, TAUCOR's
  ! verdict; per work item its k-term; the fleet whose devices hold the depths
  integer(c_int32_t), allocatable, target :: gas_nk(:), gas_fail(:), kterm(:)
  real(kr), allocatable, target :: gas_wt(:, :), gas_depths(:, :, :)
  type(c_ptr) :: gas_fleet
  integer(c_int64_t) :: gas_lay_token = 0               ! names the layer blocks sbd_fleet_gas_terms left on gas_fleet's devices
  integer(kind=8) :: tick_g0, tick_g1
  real(kind=8) :: t_gas
  integer(c_int) :: rc_gas

  call set_defaults()
  call warn_reset()
  aborted = .false.
  use_mix = .false.; gas_dev = .false.; gas_fleet = c_null_ptr; t_gas = 0

  ! ---- read ./INPUT exactly like drt.f:220-231 ----
  open(newunit=u11, file='INPUT', status='old', iostat=ios)
  if (ios == 0) then
    read(u11, input, iostat=ios)
    if (ios /= 0) then
      write(0, '(a)') 'error: namelist block $INPUT not found'
      call leave(); return
    end if
    read(u11, dinput, iostat=ios)
    close(u11)
  else
    write(*, input)
    call leave(); return
  end if

  call check_input()
  if (aborted) return
  if (iout /= 2) fmt = find_format(iout, known)
  if (iout == 2) known = .true.
  if (.not. known) call quit('this IOUT is not an output format of the hot path (1,2,5,6,7,10,11,20,21,22,23)')
  if (aborted) return
  radcalc = .false.
  if (iout /= 2) radcalc = fmt%radiance /= rad_none     ! drt.f:237-247
  if (nstr == 0) nstr = merge(min(20, nstrms), 4, radcalc)
  if (radcalc) view = new_view(iout, nphi, phi, nzen, uzen, vzen)

  ! ---- solar geometry (drt.f:275-283) ----
  if (iday /= 0 .or. isat > 0) then
    call tables_load(ok, why)
    if (.not. ok) call quit('tables not found; tried'//trim(why))
    if (aborted) return
  end if
  if (iday /= 0) then
    call solar_position(abs(iday), time, alat, alon, sza, saza, solfac)
  else if (csza /= unset) then
    sza = acos(csza)/(real(3.1415926536d0, kr)/180.)
  end if
  if (abs(sza - 90) < .01) sza = 95.
  phi0 = mod(saza - 180.0_kr + 360.0_kr, 360.0_kr)      ! drt.f:283
  if (iday < 0) then                                     ! drt.f:285-299: report the solar geometry and stop
    print '(a5,6a9)', 'day', 'time', 'lat', 'lon', 'sza', 'azm', 'solfac'
    print '(i5,6f9.3)', abs(iday), time, alat, alon, sza, saza, solfac
    if (radcalc) then
      print '(2a9)', 'phi', 'rel_az'
      do i = 1, view%nphi
        if (phi0 > 180. .and. view%phi(view%nphi) + phi0 > 360) then
          print '(2f9.3)', view%phi(i), view%phi(i) + phi0 - 360.
        else
          print '(2f9.3)', view%phi(i), view%phi(i) + phi0
        end if
      end do
    end if
    call leave(); return
  end if
  sensor = new_filter(isat, wlinf, wlsup)            ! setfilt: the sensor's response and its wavelength limits
  grid = new_grid(sensor%wlmin, sensor%wlmax, wlinc)
  if (iout == 2) kdist = 0                              ! drt.f:303
  if (kdist == -1) then                                 ! the spectral points are the bands of the k-distribution file
    call read_ck_files(sensor%wlmin, sensor%wlmax, ck)
    grid = new_grid_from_bands(ck%wl, ck%wvlo, ck%wvhi)
  end if

  if (iout == 2) then                                   ! gas optical depths only: no radiative transfer
    call fill_model()
    if (.not. covered_by_band_model(model, why)) call quit('IOUT=2: the band model does not cover this run: '//trim(why))
    if (aborted) return
    call tables_load(ok, why)
    if (.not. ok) call quit('band-model tables not found; tried'//trim(why))
    if (aborted) return
    call gas_depth_report(model, grid)
    call leave(); return
  end if

  ! ---- per-work-item optical properties: optics file if there is one, else the band model ----
  call get_environment_variable('SBD_OPTICS', path, plen, pstat)
  if (pstat /= 0 .or. plen <= 0) path = 'OPTICS.sbdrec'
  if (phase == 2) path = items_file                     ! batch mode: the work items phase 1 left in the run's directory
  inquire(file=trim(path), exist=have_file)
  if (have_file) then
    call read_optics(trim(path), recs, nrec)
    if (nrec < 1) call quit('optics file holds no work items')
    if (aborted) return
    nz = recs(1)%nlyr
    allocate(zlev(nz), plev(nz))
    call get_environment_variable('SBD_ATMOS', path, plen, pstat)
    if (pstat /= 0 .or. plen <= 0) path = 'ATMOS.sbdatm'
    if (phase == 2) path = items_file//'.atm'
    call read_atmosphere(trim(path), nz, zlev, plev, have_atm)
  else
    call fill_model()
    if (.not. covered_by_band_model(model, why)) &
      call quit('no optics file ('//trim(path)//') and the band model does not cover this run yet: '//trim(why))
    if (aborted) return
    call tables_load(ok, why)
    if (.not. ok) call quit('band-model tables not found; tried'//trim(why))
    if (aborted) return
    call viewing_cosines()
    if (phase == 1 .and. serve_mode) then
      ! a forked first phase of the resident server: a run whose work items would fill a file of more than serve_big bytes
      ! (DISORT's arguments per item: 5 KB at NSTR 16 x 33 layers) is handed back (exit code 77) -- the server makes it
      ! whole, compact form and gas terms on the device, like a process of its own would (phase 3)
      if (8.0d0*real(grid%n, 8)*3.0d0*real(max(33, abs(ngrid)), 8) &
          *real(merge(302, min(max(nstr, 4) + 2, 40) + 4, corint), 8) > serve_big) then
        flush(6)
        call sbd_px_exit_now(77_c_int)
      end if
    end if
    call system_clock(tick0, tick_rate)
    ! the batch in compact form (the scatterers per spectral point, the gas per work item; DISORT's arguments are formed
    ! on the device) whenever the run's output is the engine's: not when the work items themselves are asked for
    ! (SBD_DUMP_OPTICS, phase 1 of a batch) and not for IBCND = 1 (no solve at all).  SBD_NO_MIX=1 keeps the arrays form.
    call get_environment_variable('SBD_DUMP_OPTICS', path, plen, pstat)
    mix%want = (phase == 0 .or. phase == 3) .and. .not. (pstat == 0 .and. plen > 0) .and. ibcnd /= 1
    call get_environment_variable('SBD_NO_MIX', path, plen, pstat)
    if (pstat == 0 .and. plen > 0) mix%want = .false.
    ! ... and the gas terms themselves on the device (sbd_fleet_gas_terms: the band model then delivers one record per
    ! spectral point, the engine says how many k-terms each has).  SBD_HOST_GAS=1 keeps them on the host.
    call get_environment_variable('SBD_DUMP_MIX', path, plen, pstat)
    mix%gas_on_device = mix%want .and. .not. (pstat == 0 .and. plen > 0)
    ! ... for runs large enough to pay for it: below gas_device_min spectral points the host's gas terms cost a few
    ! milliseconds, while the device's put the runtime's bring-up and one more kernel family's code object ahead of
    ! everything else (profiles/r05_e2e_input_to_stdout.txt: 751 wavelengths 0.125 s with the gas on the host, 0.21-0.29 s
    ! on the device; 39 897 and 75 001 wavelengths: the device ahead or level).  SBD_HOST_GAS=1 / SBD_DEVICE_GAS=1 decide
    ! for any size.
    if (grid%n < gas_device_min) mix%gas_on_device = .false.
    call get_environment_variable('SBD_DEVICE_GAS', path, plen, pstat)
    if (pstat == 0 .and. plen > 0) mix%gas_on_device = mix%want
    call get_environment_variable('SBD_HOST_GAS', path, plen, pstat)
    if (pstat == 0 .and. plen > 0) mix%gas_on_device = .false.
    ! ... and with them the scatterers' part (sbd_fleet_point_terms: Rayleigh, the cloud deck, the aerosols -- the layer
    ! blocks are made on the devices and never exist on the host).  SBD_HOST_SCAT=1 keeps that part on the host.
    call get_environment_variable('SBD_HOST_SCAT', path, plen, pstat)
    mix%scat_on_device = mix%gas_on_device .and. .not. (pstat == 0 .and. plen > 0)
    if (kdist == -1) then
      call build_work_items(model, grid, umu(1:numu), phiv(1:merge(view%nphi, 0, radcalc)), recs, nrec, atm, &
                            bdtauc, bssalb, bpmom, btemper, ck, mix)
    else
      call build_work_items(model, grid, umu(1:numu), phiv(1:merge(view%nphi, 0, radcalc)), recs, nrec, atm, &
                            bdtauc, bssalb, bpmom, btemper, mixb=mix)
    end if
    use_mix = mix%ok
    gas_dev = use_mix .and. mix%gas_on_device
    scat_dev = gas_dev .and. mix%scat_dev
    ! SBD_DUMP_MIX=file: the run's batch in compact form, for inspection / tests (stream: int32 nz, channels, points,
    ! terms, family(6), items; then lay, the items' gas depths and their points, 0-based) -- and stop before the engine
    call get_environment_variable('SBD_DUMP_MIX', path, plen, pstat)
    if (pstat == 0 .and. plen > 0) then
      open(newunit=u11, file=trim(path), access='stream', form='unformatted', status='replace')
      if (use_mix) then
        write(u11) int(atm%nz, 4), int(size(mix%lay, 2), 4), int(size(mix%lay, 3), 4), int(mix%nterm, 4), &
                   int(mix%family, 4), int(nrec, 4)
        write(u11) mix%lay
        write(u11) bdtauc
        write(u11) (int(recs(i)%iwl - 1, 4), i = 1, nrec)
        ! ... and the run's gas model with the items' k-term bookkeeping (tests of the engine's gas kernel)
        write(u11) merge(1_4, 0_4, mix%gas_ok)
        if (mix%gas_ok) then
          write(u11) int(mix%kdist, 4), mix%amu_gas, mix%xo4, mix%uu, mix%z, mix%wl
          write(u11) (int(recs(i)%kd, 4), int(recs(i)%nk, 4), recs(i)%wt, i = 1, nrec)
        end if
        ! ... and the run's scatterers as sbd_scat_model takes them (tests of sbd_scat.hpp: `lay` above is what they give)
        write(u11) merge(1_4, 0_4, mix%scat_ok)
        if (mix%scat_ok) then
          write(u11) mix%atm%z, mix%atm%p, mix%atm%t, mix%xrsc
          write(u11) int(mix%ncloud_term, 4), int(mix%deck%nslot, 4), int(mix%deck%layer, 4), mix%deck%tcloud, &
                     mix%deck%lwp, mix%deck%nre
          write(u11) int(mix%load%iaer, 4), int(mix%load%nosct, 4), int(mix%load%nwl, 4), mix%load%abaer
          if (mix%load%iaer /= 0) write(u11) mix%load%wl(1:mix%load%nwl), mix%load%ext(1:mix%load%nwl), &
                                             mix%load%absb(1:mix%load%nwl), mix%load%asym(1:mix%load%nwl), mix%load%column(1:atm%nz)
          write(u11) int(mix%load%nstrat, 4), int(mix%load%jaer, 4), int(mix%load%strat_layer, 4), mix%load%taerst
        end if
      else
        write(u11) 0_4, 0_4, 0_4, 0_4, (0_4, i = 1, 6), 0_4
        write(0, '(a)') 'sbdart_amd: arrays form: '//trim(mix%why)
      end if
      close(u11)
      call leave(); return
    end if
    from_model = .true.
    call system_clock(tick1)
    tick_bm0 = tick0; tick_bm1 = tick1
    call get_environment_variable('SBD_TIMING', path, plen, pstat)
    if (pstat == 0 .and. plen > 0) write(0, '(a,i0,a,i0,a,f9.4,a)') 'sbdart_amd: band model: ', grid%n, &
      ' wavelengths, ', nrec, ' work items in ', real(tick1 - tick0, kr)/real(tick_rate, kr), ' s'
    nz = atm%nz
    allocate(zlev(nz), plev(nz))
    zlev = atm%z; plev = atm%p
    do i = 1, nrec
      recs(i)%ff = filter_value(sensor, recs(i)%wl)*recs(i)%ewcoef      ! drt.f:461 (ewcoef = 1 without a k-distribution file)
    end do
    if (.not. use_mix) call corint_history(recs, nrec, bssalb)     ! (compact form: never with the intensity corrections)
    have_atm = .true.
  end if
  nmom = maxval(recs(1:nrec)%nmom)
  call get_environment_variable('SBD_DUMP_OPTICS', path, plen, pstat)      ! the work items, for inspection / tests
  if (phase == 1 .and. serve_mode .and. from_model .and. ibcnd /= 1) then
    ! a forked first phase of the resident server whose work items would fill more than serve_mid bytes (a 751-wavelength
    ! run at NSTR 16: 10 MB written here and read back there) was a REHEARSAL: screening and band model came through
    ! without a STOP -- the server now makes the run whole itself (exit code 78 -> phase 3, compact form; what stopped
    ! nothing here stops nothing there: same INPUT, same files, same code)
    if (8.0d0*real(nrec, 8)*real(nz, 8)*real(nmom + 4, 8) > serve_mid) then
      flush(6)
      call sbd_px_exit_now(78_c_int)
    end if
  end if
  if (phase == 1) then                                  ! batch mode: hand the work items to the process that solves
    path = items_tmp; plen = len(items_tmp); pstat = 0    ! (renamed to items_file by the worker once this child has exited with code 0)
    if (have_atm) call write_atmosphere(items_file//'.atm', nz, zlev, plev)
  end if
  if (pstat == 0 .and. plen > 0) then
    if (from_model) then
      call write_optics(trim(path), recs, nrec, bdtauc, bssalb, bpmom, btemper, umu(1:numu), &
                        phiv(1:merge(view%nphi, 0, radcalc)))
    else
      call write_optics(trim(path), recs, nrec)
    end if
    call leave(); return
  end if

  ! the spectral grid of INPUT must be the one the optics were made for (wllimits, drt.f:1657-1740)
  do i = 1, nrec
    if (recs(i)%iwl < 1 .or. recs(i)%iwl > grid%n) call quit('optics record outside the spectral grid of INPUT')
    if (aborted) return
    call grid%band(recs(i)%iwl - 1, wl, wvlo, wvhi)
    if (abs(wl - recs(i)%wl) > 1e-12_kr*wl .or. abs(wvlo - recs(i)%wvnmlo) > 1e-9_kr*wvlo .or. &
        abs(wvhi - recs(i)%wvnmhi) > 1e-9_kr*wvhi) call quit('optics record disagrees with the wavelength grid of INPUT')
    if (aborted) return
    if (recs(i)%nlyr /= nz .or. recs(i)%nmom > nmom .or. recs(i)%nmom < min(recs(i)%nstr, nmom)) &
      call quit('optics records differ in NLYR/NMOM')        ! (fewer moments: after CORINT went off, corint_history)
    if (aborted) return
  end do

  ! ---- output levels: the computational levels nearest to ZOUT (drt.f:368-381); level 1 = top ----
  default_zout = zout(1) == 0._kr .and. zout(2) == 100._kr
  if (have_atm) then
    nbot = nz - nearest_level(zlev, abs(zout(1))) + 2
    ntop = nz - nearest_level(zlev, abs(zout(2))) + 2
    if (ntop == 2) ntop = 1
  else
    if (.not. default_zout) call quit('ZOUT needs the level altitudes: no atmosphere file (SBD_ATMOS / ATMOS.sbdatm)')
    if (aborted) return
    if (fmt%code == 7 .or. fmt%code == 11 .or. fmt%code == 22) &
      call quit('this IOUT prints altitudes/pressures: no atmosphere file (SBD_ATMOS / ATMOS.sbdatm)')
    if (aborted) return
    ntop = 1; nbot = nz + 1
  end if
  if (fmt%profile .or. fmt%radiance == rad_levels) then
    nlev = nz + 1
    allocate(level_out(nlev))
    level_out = (/(i - 1, i = 1, nlev)/)
    lev_top = ntop; lev_bot = nbot
  else
    nlev = 2
    allocate(level_out(2))
    level_out = (/ntop - 1, nbot - 1/)
    lev_top = 1; lev_bot = 2
  end if

  if (.not. allocated(umu)) call viewing_cosines()

  ! ---- the per-run arguments every engine of the run is created with (drt.f:330-335) ----
  allocate(temper(0:nz))
  if (from_model) then
    temper = btemper
  else
    temper = recs(1)%a%temper
  end if
  if (btemp < 0._kr) btemp = recs(1)%btemp          ! drt.f:334-335 defaults come with the profile
  if (ttemp < 0._kr) ttemp = recs(1)%ttemp

  ! ---- the gas terms on the device: so far the band model delivered ONE record per spectral point; the engine
  !      evaluates gasset / depthscl for all of them at once and keeps the depths, the host gets the number of
  !      k-terms per point and their weights and makes the work items of them ----
  if (gas_dev) then
    call system_clock(tick_g0)
    npart = 3*nrec                                     ! (an upper bound for the engines' batch size)
    gas_fleet = fleet_for(nstr, .false., rc_gas)       ! (any engine of the run will do; SBD_E_RETRY_NSTR creates one too)
    if (aborted) return
    call gas_terms_on(gas_fleet, .false.)
    if (aborted) return
    if (any(gas_fail /= 0)) then
      ! TAUCOR's iteration failed somewhere: the reference prints its operands and stops there (taugas.f:7684-7690) --
      ! the host's own gas terms reproduce that to the letter.  (Not by starting this executable again in place: an
      ! exec of a process that holds a GPU context, followed by the new image opening the device under the same PID,
      ! took the GPU box down twice in round 6.)
      call quit('the slant-path correction did not converge at some wavelength: run with SBD_HOST_GAS=1 for the reference''s report')
      return
    end if
    call system_clock(tick_g1)
    t_gas = real(tick_g1 - tick_g0, 8)/real(tick_rate, 8)
    call expand_work_items(recs, nrec, int(gas_nk), gas_wt)
  end if

  ! ---- batch arrays (row-major by work item == Fortran's first index fastest).  Items the filter
  !      function removes (ff = 0) are not solved (drt.f:461-462); items with a beam come first so
  !      that an NSTR retry (below) re-solves one contiguous part, and among them first those whose call
  !      still asks for the intensity corrections (flag 16: corint_history) ----
  call system_clock(tick0, tick_rate)
  allocate(order(nrec), where_solved(nrec))
  where_solved = 0                                   ! batch position of record i, 0 = not solved
  ! (the part of every record first -- 0 not solved, 1 beam with corrections, 2 beam, 3 no beam --, all records at once:
  !  the records are hundreds of bytes apart, three serial walks over 200 000 of them cost tens of milliseconds)
  allocate(part_of(nrec))
  !$omp parallel do schedule(static) num_threads(max(1, min(omp_get_max_threads(), nrec/4096 + 1, 16)))
  do i = 1, nrec
    if (recs(i)%ff == 0._kr) then
      part_of(i) = 0
    else if (.not. recs(i)%fbeam > 0._kr) then
      part_of(i) = 3
    else if (corint .and. radcalc .and. iand(recs(i)%flags, 16) == 0) then
      part_of(i) = 2
    else
      part_of(i) = 1
    end if
  end do
  !$omp end parallel do
  npart = 0
  do pass = 1, 3
    do i = 1, nrec
      if (part_of(i) /= pass) cycle
      npart = npart + 1
      order(npart) = i
      where_solved(i) = npart
    end do
    if (pass == 1) ncorr = npart
    if (pass == 2) nbeam = npart
  end do
  ! (the band model's arrays are already the batch when every item is solved and all are of one kind)
  in_place = from_model .and. npart == nrec .and. (ncorr == nrec .or. nbeam - ncorr == nrec .or. nbeam == 0) .and. .not. gas_dev
  allocate(wvnmlo(nrec), wvnmhi(nrec), fbeam(nrec), albedo(nrec), plank(nrec), status(nrec), weight(nrec), kterm(nrec))
  allocate(bitem(4, nrec))
  allocate(pmom_row(nrec))
  if (from_model) then
    call move_alloc(bpmom, pmom)                       ! one block of moments per wavelength: pmom_row picks it per item
  else
    allocate(pmom(0:nmom, nz, nrec))
  end if
  ! (compact form: `dtauc` holds the GAS depth of every item -- sbd_mix_in%dtaug --, `ssalb` and `pmom` are not formed
  !  on the host at all)
  if (in_place) then
    call move_alloc(bdtauc, dtauc); call move_alloc(bssalb, ssalb)
  else
    allocate(dtauc(nz, merge(1, nrec, gas_dev)), ssalb(nz, merge(1, nrec, use_mix)))      ! (gas on the device: no depths here at all)
  end if
  status = 0
  !$omp parallel do schedule(static) private(i) num_threads(max(1, min(omp_get_max_threads(), npart/4096 + 1, 16)))
  do ip = 1, npart
    i = order(ip)
    if (.not. in_place) then
      if (gas_dev) then
        continue
      else if (from_model) then
        dtauc(:, ip) = bdtauc(:, i)
        if (.not. use_mix) ssalb(:, ip) = bssalb(:, i)
      else
        dtauc(:, ip) = recs(i)%a%dtauc; ssalb(:, ip) = recs(i)%a%ssalb
        pmom(:, :, ip) = 0                             ! (a run whose CORINT went off holds shorter moment arrays later)
        pmom(0:recs(i)%nmom, :, ip) = recs(i)%a%pmom
      end if
    end if
    wvnmlo(ip) = recs(i)%wvnmlo; wvnmhi(ip) = recs(i)%wvnmhi
    fbeam(ip) = recs(i)%fbeam; albedo(ip) = recs(i)%albedo
    plank(ip) = int(iand(recs(i)%flags, 1), c_int8_t)
    bitem(:, ip) = recs(i)%bitem
    pmom_row(ip) = recs(i)%iwl - 1                     ! (0-based block of the item's wavelength; used when from_model)
    kterm(ip) = recs(i)%kd - 1
    weight(ip) = recs(i)%wt*recs(i)%ff                 ! dwt of stdout1 (drt.f:964)
  end do
  !$omp end parallel do
  if (use_mix) then                                    ! (a point's scalars: any of its items carries them)
    allocate(pt_lo(grid%n), pt_hi(grid%n), pt_fb(grid%n), pt_al(grid%n), pt_pl(grid%n))
    pt_lo = 0; pt_hi = 0; pt_fb = 0; pt_al = 0; pt_pl = 0
    !$omp parallel do schedule(static) private(j) num_threads(max(1, min(omp_get_max_threads(), nrec/4096 + 1, 16)))
    do i = 1, nrec
      if (recs(i)%kd /= 1) cycle                         ! (a point's first k-term speaks for the point)
      j = recs(i)%iwl
      pt_lo(j) = recs(i)%wvnmlo; pt_hi(j) = recs(i)%wvnmhi; pt_fb(j) = recs(i)%fbeam; pt_al(j) = recs(i)%albedo
      pt_pl(j) = int(iand(recs(i)%flags, 1), c_int8_t)
    end do
    !$omp end parallel do
  end if
  allocate(flux(nlev, SBD_NFLUX, nrec), acc_flux(nlev, SBD_NFLUX))
  if (radcalc) then
    allocate(uu(numu, nlev, view%nphi, nrec), acc_uu(numu, nlev, view%nphi))
  else
    allocate(uu(1, 1, 1, 1), acc_uu(1, 1, 1))
  end if
  flux = 0; uu = 0; acc_flux = 0; acc_uu = 0

  ! ---- the wavelength loop: the beam items with the reference's NSTR "dithering" (a beam angle that
  !      coincides with a quadrature angle makes DISORT ask for another stream count, drt.f:536-555),
  !      the beamless items with the stream count as given ----
  call system_clock(tick1)
  ! (IBCND = 1 in &DINPUT: the reference hands it to DISORT, which then returns the medium's albedo and
  !  transmissivity in two arguments SBDART never looks at and leaves every flux and intensity at zero
  !  (disort.f:545-556) -- the run prints zeros; so does this one, without a solve.  The engine offers the
  !  mode itself through sbd_run_cfg::ibcnd.)
  call get_environment_variable('SBD_ORDERED_SUMS', path, plen, pstat)
  items_wanted = fmt%per_point .or. (pstat == 0 .and. plen > 0 .and. path(1:1) /= '0')
  if (ibcnd /= 1) then
    call solve_part(1, ncorr, .true., corint)
    if (.not. aborted) call solve_part(ncorr + 1, nbeam, .true., .false.)
    if (.not. aborted) call solve_part(nbeam + 1, npart, .false., .false.)
    ! (an item CHEKIN refused: its report below wants the item's arguments -- the gas depths come back from the
    !  devices while their engines still exist)
    if (gas_dev .and. .not. aborted) then
      if (any(iand(status(1:npart), SBD_ST_ERR_INPUT) /= 0)) call gas_terms_on(gas_fleet, .true.)
    end if
    if (phase == 0) call release_fleets()               ! (a batch keeps its fleets for the runs that follow)
    if (aborted) return
  end if
  call system_clock(tick2)
  t_engine = t_engine + real(tick2 - tick1, 8)/real(tick_rate, 8)
  call get_environment_variable('SBD_TIMING', path, plen, pstat)
  if (phase == 2) pstat = 1                             ! (a batch reports its totals once, at the end)
  if (pstat == 0 .and. plen > 0) write(0, '(a,f9.4,a,i0,a,f9.4,a)') 'sbdart_amd: batch assembly ', &
    real(tick1 - tick0, kr)/real(tick_rate, kr), ' s; engine (create + H2D + solve + D2H), ', npart, ' solves: ', &
    real(tick2 - tick1, kr)/real(tick_rate, kr), ' s'

  ! ---- warnings / fatals the reference raises through errmsg ----
  stall = 0
  do i = 1, npart
    stall = ior(stall, status(i))
  end do
  ! (input errors: the reference stops INSIDE the DISORT call that finds them -- what it printed for the wavelengths
  !  before stays on stdout.  With a bidirectional surface CHEKIN also reports the offending cosines there; that case
  !  is replayed in order below, at the record where it happens)
  fatal_at = 0
  if (iand(stall, SBD_ST_ERR_INPUT) /= 0) then
    do i = 1, nrec
      ip = where_solved(i)
      if (ip > 0) then
        if (iand(status(ip), SBD_ST_ERR_INPUT) /= 0) then
          fatal_at = i
          exit
        end if
      end if
    end do
    if (fatal_at == 0) then
      call warn_file(0, 'DISORT--input and/or dimension errors', phase == 0)
      call leave(); return
    end if
    ! The reference stops inside the call of record fatal_at: the calls after it never happen, and neither do their
    ! warnings -- the batch solved them all, so the warnings are collected again from the records before the stop only
    ! (the last differing warning-file set of the end-to-end fuzz: errmsg 4 from an item the reference never reached).
    stall = 0
    do i = 1, fatal_at
      ip = where_solved(i)
      if (ip > 0) stall = ior(stall, iand(status(ip), not(SBD_ST_ERR_INPUT)))
    end do
  end if
  if (iand(stall, SBD_ST_ERR_EIGEN) /= 0) then
    call warn_file(0, 'ASYMTX--convergence problems', phase == 0)
    call leave(); return
  end if
  if (iand(stall, SBD_ST_WARN_SOLVE0) /= 0) call warn_file(2, 'SOLVE0--SGBCO says matrix near singular')
  if (iand(stall, SBD_ST_WARN_UPBEAM) /= 0) call warn_file(3, 'UPBEAM--SGECO says matrix near singular')
  if (iand(stall, SBD_ST_WARN_UPISOT) /= 0) call warn_file(4, 'UPISOT--SGECO says matrix near singular')
  if (iand(stall, SBD_ST_WARN_PLKCONV) /= 0) call warn_file(9, 'PLKAVG--Simpson rule didnt converge')
  if (iand(stall, SBD_ST_WARN_PLKAVG) /= 0) call warn_file(10, 'PLKAVG--returns zero; possible underflow')
  if (any(plank(1:npart) /= 0)) then                ! CHEKIN warning 6 (disort.f:5145-5152)
    do i = 1, nz
      if (abs(temper(i) - temper(i - 1)) > 10._kr) then
        call warn_file(6, 'CHEKIN--vertical temperature step may be too large for good accuracy')
        exit
      end if
    end do
  end if
  ! CHEKIN warning 5 (disort.f:4939-4941): looked at in every call whose CORINT argument is still true on entry (flag 16,
  ! corint_history) -- the moments of THAT call, its highest one.  (Until round 4: any moment block of the run, also
  ! of wavelengths never solved or solved after CORINT had gone off; the end-to-end fuzz found the two runs that differ.)
  if (corint .and. npart > 0) then
    do i = 1, nrec
      ip = where_solved(i)
      if (ip == 0 .or. iand(recs(i)%flags, 16) == 0 .or. recs(i)%nmom <= 10) cycle
      j = merge(int(pmom_row(ip)) + 1, ip, from_model)
      if (any(pmom(recs(i)%nmom, :, j) > real(1.e-3, kr))) then
        call warn_file(5, 'CHEKIN-- phase function not sufficiently resolved for use with corint=.true.')
        exit
      end if
    end do
  end if
  ! CHEKIN warning 7 (disort.f:5154-5158, 5169): every beam call whose CORINT argument is false -- the namelist's
  ! value, or the one DISORT switched off in place at an earlier beamless call (corint_history: items ncorr+1..nbeam)
  ! -- and in which something scatters (YESSCT > 0, disort.f:5163-5166: the fuzz's Rayleigh-free clear-sky runs)
  if (radcalc .and. nbeam > 0 .and. (.not. corint .or. nbeam > ncorr)) then
    do ip = merge(ncorr + 1, 1, corint), nbeam
      if (use_mix) then                                  ! (SSALB = tsc / DTAUC where DTAUC > tiny: positive iff tsc is)
        if (scat_dev .and. size(mix%lay, 3) == 0) then   ! (the blocks are on the devices: this point's, from the same source)
          known = any(host_block_of(int(pmom_row(ip)) + 1) > tiny(1._kr))
        else
          known = any(mix%lay(:, 4, int(pmom_row(ip)) + 1) > tiny(1._kr))
        end if
      else
        known = sum(ssalb(:, ip)) > 0._kr
      end if
      if (known) then
        call warn_file(7, 'CHEKIN--intensity correction is off; intensities may be less accurate')
        exit
      end if
    end do
  end if

  ! ---- output ----
  call sums_init(sums, fmt, nz, view%nzen, view%nphi)
  if (kdist == -1) then                                  ! (gasinit counts the file's points, taugas.f:7362-7364)
    call write_banner(fmt, count(recs(1:nrec)%ib == 1 .and. recs(1:nrec)%kd == recs(1:nrec)%nk), nz)
  else
    call write_banner(fmt, grid%n, nz)
  end if
  call get_environment_variable('SBD_SUBBAND_WIDTHS', path, plen, pstat)
  sum_widths = pstat == 0 .and. plen >= 3 .and. path(1:3) == 'sum'
  if (fmt%per_point) then
    ! one record per spectral point: its k-terms are consecutive records (kd = 1..nk) -- and, with a
    ! k-distribution file, so are its sub-bands (ib = nb..1): the point is complete at kd = nk, ib = 1 and the
    ! wavelength printed is the last sub-band's (drt.f:967-994).  The widths: stdout1 means to add dwl*ff and dwl up
    ! over the sub-bands (drt.f:984-987), but keeps the two sums in locals it never SAVEs (drt.f:957-959) -- the
    ! reference as compiled here (amdflang -O2) starts them from zero at every call, so what it prints is the LAST
    ! sub-band's width alone under the fluxes of all of them.  That observable behaviour is reproduced (the live
    ! comparison of tests/test_fortran_host.py pins it); SBD_SUBBAND_WIDTHS=sum selects the intended sums.
    i0 = 1
    do while (i0 <= nrec)
      i1 = i0
      do while (i1 < nrec)
        if (recs(i1)%kd == recs(i1)%nk .and. recs(i1)%ib == 1) exit
        i1 = i1 + 1
      end do
      if (fatal_at >= i0 .and. fatal_at <= i1) then
        call input_stop(fatal_at)
        call leave(); return
      end if
      call sums_clear(sums)
      sums%width_eq = 0; sums%width_full = 0
      do i = i0, i1
        ip = where_solved(i)
        if (ip > 0) call sums_add_item(sums, fmt, weight(ip), flux(:, 1:3, ip), lev_top, lev_bot, &
                                       uu(:, :, :, merge(ip, 1, radcalc)), view%uzen)
        if (recs(i)%kd == recs(i)%nk .and. (sum_widths .or. i == i1)) then
          dwl = 10000._kr/recs(i)%wvnmlo - 10000._kr/recs(i)%wvnmhi    ! drt.f:438
          sums%width_eq = sums%width_eq + dwl*recs(i)%ff
          sums%width_full = sums%width_full + dwl
        end if
      end do
      call write_point_record(sums, fmt, recs(i1)%wl, zlev, view%phi, view%uzen)
      i0 = i1 + 1
    end do
  else
    ! one record per run: the engine's reduced sums; the equivalent width from the last k-term of each point
    ! (the engine adds per part -- corrected beam items, other beam items, beamless items -- and per device shard:
    !  the association of the sums, hence their last bits, follows the part split and the number of GPUs.
    !  SBD_ORDERED_SUMS=1 adds the per-item outputs here instead, in the reference's wavelength order
    !  (drt.f:964-1054): bit-reproducible on any number of devices)
    if (fatal_at > 0) then
      call input_stop(fatal_at)
      call leave(); return
    end if
    call get_environment_variable('SBD_ORDERED_SUMS', path, plen, pstat)
    if (pstat == 0 .and. plen > 0 .and. path(1:1) /= '0') then
      do i = 1, nrec
        ip = where_solved(i)
        if (ip > 0) call sums_add_item(sums, fmt, weight(ip), flux(:, 1:3, ip), lev_top, lev_bot, &
                                       uu(:, :, :, merge(ip, 1, radcalc)), view%uzen)
      end do
    else
      call sums_from_engine(sums, fmt, acc_flux, acc_uu, lev_top, lev_bot, view%uzen)
    end if
    do i = 1, nrec
      if (recs(i)%kd == recs(i)%nk) then
        dwl = 10000._kr/recs(i)%wvnmlo - 10000._kr/recs(i)%wvnmhi
        sums%width_eq = sums%width_eq + dwl*recs(i)%ff
      end if
    end do
    call write_run_record(sums, fmt, sensor%wlmin, sensor%wlmax, zlev, plev, view%phi, view%uzen)   ! (wl1, wl2 of setfilt)
  end if
  ! SBD_TIMING: the run's account in one line (seconds inside the process, counted from the program's first statement;
  ! compact = 1 compact form, 2 with the gas terms on the device, 3 with the layer blocks made there as well):
  ! setup = namelist, screening, tables, grid; band_model; assembly = batch arrays; engine = fleet create (of which
  ! engine_create) + H2D + kernels + D2H / reduce; output = warnings and writers; total = first statement -> here
  call get_environment_variable('SBD_TIMING', path, plen, pstat)
  if (pstat == 0 .and. plen > 0 .and. phase == 0 .and. from_model .and. tick_program >= 0) then
    flush(6)
    call system_clock(tick_out)
    write(0, '(a,i0,a,i0,a,i0,8(a,f0.4))') 'sbdart_amd: timing nwl=', grid%n, ' items=', npart, ' compact=', merge(1, 0, use_mix) + merge(1, 0, gas_dev) + merge(1, 0, scat_dev), &
      ' setup=', real(tick_bm0 - tick_program, 8)/real(tick_rate, 8), ' band_model=', real(tick_bm1 - tick_bm0, 8)/real(tick_rate, 8), &
      ' assembly=', real(tick1 - tick0, 8)/real(tick_rate, 8) + real(tick0 - tick_bm1, 8)/real(tick_rate, 8), &
      ' gas_device=', t_gas, ' engine=', real(tick2 - tick1, 8)/real(tick_rate, 8), ' engine_create=', t_create, &
      ' output=', real(tick_out - tick2, 8)/real(tick_rate, 8), ' total=', real(tick_out - tick_program, 8)/real(tick_rate, 8)
  end if
  call get_environment_variable('SBD_SUMS_FILE', path, plen, pstat)   ! full-precision sums for parity tests
  if (pstat == 0 .and. plen > 0) then
    open(newunit=u11, file=trim(path), status='replace', form='formatted')
    write(u11, '(6es25.16)') sums%down(1), sums%up(1), sums%direct(1), sums%down(2), sums%up(2), sums%direct(2)
    close(u11)
  end if

contains

  ! the end of a run before its last line: a single run stops the process as the reference does; a run of a batch
  ! only marks itself finished and the caller RETURNs (every call is followed by one)
  subroutine leave()
    aborted = .true.
    if (phase == 0) stop
  end subroutine

  ! a condition this host cannot continue from.  One run per process (phase 0, and a batch's phase 1 in its own
  ! child): the message and a non-zero exit, like the reference's STOPs.  Phase 2 of a batch runs in the process that
  ! owns the GPU and serves the runs that follow: the message goes to the run's SBDART.stderr and only this run ends
  ! (every call site returns on `aborted`).
  subroutine quit(msg)
    character(len=*), intent(in) :: msg
    integer :: ue, ios2
    if (phase == 0 .or. phase == 1) call fatal(msg)
    aborted = .true.
    ios2 = 1
    if (.not. serve_mode) open(newunit=ue, file=stderr_file, position='append', action='write', iostat=ios2)
    if (ios2 == 0) then
      write(ue, '(a)') 'sbdart_amd: '//msg
      close(ue)
    else
      write(0, '(a)') 'sbdart_amd: '//msg
    end if
  end subroutine

  subroutine set_defaults()                            ! drt.f:144-198: the namelist's defaults
    idatm = 4; isat = 0; nf = 2; iday = 0; isalb = 0; krhclr = 0; jaer = 0; iaer = 0
    nothrm = -1; nosct = 0; kdist = 3; ngrid = 0; idb = 0; iout = 10; nstr = 0; nzen = 0
    nphi = 0; imomc = 3; imoma = 3; ibcnd = 0; ipth = 0
    amix = unset; wlinf = real(.55, kr); wlsup = real(.55, kr); wlinc = 0; sza = 0; csza = unset; solfac = 1
    time = 16; alat = real(-64.767, kr); alon = real(-64.067, kr); zpres = unset; pbar = unset; sclh2o = unset
    uw = unset; uo3 = unset; o3trp = unset; ztrp = 0; xrsc = 1; xn2 = unset; xo2 = unset; xco2 = unset; xch4 = unset
    xn2o = unset; xco = unset; xno2 = unset; xso2 = unset; xnh3 = unset; xno = unset; xhno3 = unset; xo4 = 1
    albcon = 0; sc = huge(0.); zcloud = 0; tcloud = 0; lwp = 0; nre = 8
    rhcld = unset; zaer = 0; taerst = 0; vis = unset; rhaer = unset; tbaer = unset
    wlbaer = unset; qbaer = unset; abaer = 0; wbaer = unset; gbaer = unset
    pmaer = unset; zbaer = unset; dbaer = unset; zgrid1 = 1; zgrid2 = 30
    zout = (/0._kr, 100._kr/); temis = 0; uzen = unset; vzen = 90; phi = unset
    saza = 180; ttemp = unset; btemp = unset; phi0 = 0; fisot = 0
    prnt = .false.; corint = .false.; spowder = .false.
    from_model = .false.
  end subroutine

  subroutine fill_model()                              ! the &INPUT variables the band model reads
    model%idatm = idatm; model%nf = nf; model%isalb = isalb; model%kdist = kdist; model%nothrm = nothrm
    model%ngrid = ngrid; model%nstr = nstr
    model%aer%iaer = iaer; model%aer%jaer = jaer; model%aer%imoma = imoma; model%aer%nosct = nosct
    model%aer%zaer = zaer; model%aer%taerst = taerst; model%aer%vis = vis; model%aer%tbaer = tbaer
    model%aer%abaer = abaer; model%aer%rhaer = rhaer; model%aer%wlbaer = wlbaer; model%aer%qbaer = qbaer
    model%aer%wbaer = wbaer; model%aer%gbaer = gbaer; model%aer%zbaer = zbaer; model%aer%dbaer = dbaer
    model%aer%pmaer = pmaer
    model%amix = amix; model%sza = sza; model%solfac = solfac; model%albcon = albcon; model%xrsc = xrsc
    model%zpres = zpres; model%pbar = pbar; model%sclh2o = sclh2o; model%uw = uw; model%uo3 = uo3
    model%o3trp = o3trp; model%ztrp = ztrp
    model%xgas = (/xn2, xo2, xco2, xch4, xn2o, xco, xno2, xso2, xnh3, xno, xhno3/)
    model%xo4 = xo4; model%btemp = btemp; model%ttemp = ttemp; model%temis = temis; model%fisot = fisot
    model%phi0 = phi0
    model%zcloud = zcloud; model%tcloud = tcloud; model%lwp = lwp; model%nre = nre; model%rhcld = rhcld
    model%imomc = imomc; model%krhclr = krhclr
    where (sc == huge(0.)) sc = (/1._kr, 0._kr, 0._kr, 0._kr, 0._kr/)          ! drt.f:249-262
    model%sc = sc
    model%zgrid1 = zgrid1; model%zgrid2 = zgrid2
    model%spowder = spowder; model%radiance = radcalc; model%corint = corint
  end subroutine

  subroutine viewing_cosines()                         ! drt.f:391-403: ascending cosines, never exactly 0
    numu = 0
    if (radcalc) then
      numu = view%nzen
      allocate(umu(numu), phiv(view%nphi))
      do j = 1, numu
        umu(j) = min(1._kr, max(cos(view%uzen(numu + 1 - j)*(real(3.1415926536d0, kr)/180._kr)), -1._kr))
        if (umu(j) == 0._kr) umu(j) = merge(-real(.0001, kr), real(.0001, kr), j == numu)
      end do
      phiv = view%phi(1:view%nphi)
    else
      allocate(umu(1), phiv(1))
    end if
  end subroutine

  ! The reference's input screening (chkin, drt.f:568-728): out-of-range namelist values are reported
  ! with its messages and stop the run; two combinations only warn (errmsg 16/17).  One rule per line:
  ! the condition that makes the value unacceptable, the name and the range text that are printed.
  subroutine check_input()
    nbad = 0
    if (iaer == 0 .and. (vis /= unset .or. tbaer /= unset)) call warn_file(16, 'CHKIN--IAER=0, though VIS or TBAER set')
    if (corint .and. .not. any(iout == (/5, 6, 20, 21, 22, 23/))) &
      call warn_file(17, 'CHKIN--CORINT=t, but flux output selected')
    call rule(idatm < -6 .or. idatm > 6, 'idatm', '[-6,6]', iv('idatm=', (/idatm/)))
    call rule(wlinf < real(0.199, kr), 'wlinf', '[0.2,-]', rv('wlinf=', (/wlinf/)))
    if (isat <= -2) then
      call rule(wlsup < 0._kr .or. wlsup >= wlinf, 'wlsup', '[0,wlinf]', rv('wlsup=', (/wlsup/)))
    else
      call rule(wlsup < wlinf .or. wlsup > 100._kr, 'wlsup', '[wlinf,100]', rv('wlsup=', (/wlsup/)))
    end if
    call rule(isat < -4 .or. isat > 29, 'isat', '[-4,29]', iv('isat=', (/isat/)))
    call rule(solfac < 0._kr, 'solfac', '[0,inf]', rv('solfac=', (/solfac/)))
    call rule(minval(zcloud) < -100._kr .or. maxval(zcloud) > 100._kr, 'zcloud', '[-100,100]', rv('zcloud=', zcloud))
    call rule((minval(abs(nre)) < 2._kr .or. maxval(abs(nre)) > 128._kr) .and. nre(1) /= 0._kr, 'nre', '[2,128]', rv('nre', nre))
    if (any(tcloud == 0._kr .and. zcloud < 0._kr)) then
      print *, 'CHKIN --- Error detected in input'
      print *, 'TCLOUD(k)=0 when ZCLOUD(k)<0'
      nbad = nbad + 1
    end if
    call rule(minval(lwp) < 0._kr, 'lwp', '[0,inf]', rv('lwp=', lwp))
    call rule(maxval(abs(zaer)) > 100._kr, 'zaer', '[-100,100]', rv('zaer=', zaer))
    call rule(minval(taerst) < 0._kr, 'taerst', '[0,inf]', rv('taerst', taerst))
    call rule(minval(jaer) < 0 .or. maxval(jaer) > 4, 'jaer', '[0,4]', iv('jaer', jaer))
    call rule(nf < -2 .or. nf > 3, 'nf', '[-2,3]', iv('nf', (/nf/)))
    call rule(iaer < -1 .or. iaer > 5, 'iaer', '[-1,5]', iv('iaer', (/iaer/)))
    call rule(.not. any(isalb == (/-7, -8, -9, -1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10/)), 'isalb', &
              '[-7,-8,-9,-1,0,1,2,3,4,5,6,7,8,9,10]', iv('isalb', (/isalb/)))
    call rule(isalb == 0 .and. albcon < 0._kr, 'albcon', '[0,inf]', rv('albedo set by albcon', (/albcon/)))
    call rule(minval(zout) < 0._kr .or. maxval(zout) > 100._kr, 'zout', '[0,100]', rv('zout', zout))
    call rule(.not. any(iout == (/1, 2, 5, 6, 7, 10, 11, 20, 21, 22, 23/)), 'iout', '[1,2,5,6,7,10,11,20,21,22,23]', iv('iout', (/iout/)))
    call rule(nphi < 0 .or. nphi > nstrms, 'nphi', '[0,nstrms]', iv('nphi', (/nphi/)))
    if (any(iout == (/20, 21, 22, 23/)) .and. nzen == 0 .and. all(uzen == unset) .and. all(vzen == 90._kr)) then
      write(*, '(1x,a,i2,a,a)') 'iout =', iout, ' implies radiance calculation, ', 'but nzen=0 produces no radiance output'
      nbad = nbad + 1
    end if
    if (zpres /= unset .and. pbar /= unset) then
      write(*, '(1x,a)') 'set zpres or pbar but not both'
      nbad = nbad + 1
    end if
    if (any(tcloud /= 0._kr) .and. any(lwp /= 0._kr)) then
      write(*, *) 'set TCLOUD or LWP, but not both'
      nbad = nbad + 1
    end if
    ! IDB(1:9): the reference's diagnostic prints (drt.f:235 helper, 360-366 saturation and absorber amounts, 412
    ! chkprn, 429-434 k-distribution terms, 471-485 surface albedo, 507-514 cloud and aerosol tables, 533-534 depthscl)
    ! replace the run's output: with any of them set there is no banner (drt.f:327) and the wavelength loop prints the
    ! table and skips DISORT (drt.f:515, 534).  They are listings of the band model's intermediate arrays, not outputs of
    ! the hot path: refused by name rather than ignored (an INPUT that sets one would print something else here).
    do j = 1, ndb
      if (idb(j) == 0) cycle
      write(*, '(1x,a,i0,a,i0,a)') 'idb(', j, ')=', idb(j), ': the diagnostic listings of the reference (idb, drt.f:235-534) '// &
        'are not produced by sbdart_amd; unset idb to run'
      nbad = nbad + 1
    end do
    if (nbad > 0) call leave()
  end subroutine

  ! (helpers of check_input; flang takes no internal procedure inside an internal procedure of a module procedure)
  subroutine rule(violated, name, range, echo)
    logical, intent(in) :: violated
    character(len=*), intent(in) :: name, range, echo
    if (.not. violated) return
    if (nbad == 0) print '(a)', 'CHKIN --- Errors detected in INPUT'
    print '(/5x,4a)', 'Input parameter ', name, ' not within ', range
    print '(a)', echo
    nbad = nbad + 1
  end subroutine
  function iv(label, v) result(t)                   ! "label value(s)" as list-directed output prints it
    character(len=*), intent(in) :: label
    integer, intent(in) :: v(:)
    character(len=:), allocatable :: t
    character(len=512) :: buf
    write(buf, *) label, v
    t = trim(buf)
  end function
  function rv(label, v) result(t)
    character(len=*), intent(in) :: label
    real(kr), intent(in) :: v(:)
    character(len=:), allocatable :: t
    character(len=2048) :: buf
    write(buf, *) label, v
    t = trim(buf)
  end function

  ! index of the level nearest to altitude zq in the bottom-up altitudes (ties: the lower level)
  integer function nearest_level(z, zq) result(k)
    real(kr), intent(in) :: z(:), zq
    integer :: m(1)
    m = minloc(abs(z - zq))
    k = m(1)
  end function

  ! The GPUs of the run.  Default: device 0 -- one process, one GPU, sums on that device.  SBD_DEVICES=all takes
  ! every visible device (contiguous spectral shards, one RCCL reduce of the sums); SBD_DEVICES=0,2,3 a list.
  subroutine pick_devices()
    character(len=256) :: txt
    integer :: tlen, tstat, k, ios, pos, nxt
    if (allocated(devices)) return
    call get_environment_variable('SBD_DEVICES', txt, tlen, tstat)
    if (tstat /= 0 .or. tlen <= 0) then
      allocate(devices(1)); devices(1) = 0
      return
    end if
    if (trim(adjustl(txt(1:tlen))) == 'all' .or. trim(adjustl(txt(1:tlen))) == 'ALL') then
      allocate(devices(0))                             ! (zero-length: sbd_fleet_create's "every visible device")
      return
    end if
    k = 1
    do pos = 1, tlen
      if (txt(pos:pos) == ',') k = k + 1
    end do
    allocate(devices(k))
    pos = 1
    do k = 1, size(devices)
      nxt = index(txt(pos:tlen), ',')
      if (nxt == 0) nxt = tlen - pos + 2
      read(txt(pos:pos + nxt - 2), *, iostat=ios) devices(k)
      if (ios /= 0) call quit('SBD_DEVICES: expected "all" or a comma-separated list of device ordinals')
      if (aborted) return
      pos = pos + nxt
    end do
  end subroutine

  ! the fleet for stream count ns with / without the intensity corrections: created once and found again by
  ! EVERYTHING sbd_fleet_create reads (engines, workspaces and the communicator are the expensive part of a small
  ! run) -- by the parts of a run, and in a batch by the runs that follow; rc as sbd_fleet_create returned it
  function fleet_for(ns, corr, rc) result(fl)
    integer, intent(in) :: ns
    logical, intent(in) :: corr
    integer(c_int), intent(out) :: rc
    type(c_ptr) :: fl
    type(sbd_run_cfg) :: cfg
    real(kr), allocatable :: key(:)
    integer :: k, oldest
    key = (/real(ns, kr), merge(1._kr, 0._kr, corr), real(nz, kr), real(nmom, kr), merge(1._kr, 0._kr, radcalc), &
            real(recs(1)%ibdrf, kr), real(numu, kr), real(merge(view%nphi, 0, radcalc), kr), real(nlev, kr), &
            recs(1)%umu0, phi0, fisot, btemp, ttemp, temis, recs(1)%bpar, temper, umu(1:max(numu, 0)), &
            phiv(1:merge(view%nphi, 0, radcalc)), real(level_out, kr)/)
    clock = clock + 1
    do k = 1, nslot
      if (size(slots(k)%key) /= size(key) .or. slots(k)%max_batch < npart) cycle
      if (all(slots(k)%key == key)) then
        fl = slots(k)%fleet; rc = slots(k)%rc; slots(k)%age = clock
        return
      end if
    end do
    call pick_devices()
    if (aborted) then
      fl = c_null_ptr; rc = -1_c_int; return
    end if
    cfg%abi_version = SBD_ABI_VER
    cfg%nlyr = nz; cfg%nstr = ns; cfg%nmom = nmom
    cfg%onlyfl = merge(0, 1, radcalc); cfg%usrang = merge(1, 0, radcalc)
    cfg%lamber = merge(1, 0, recs(1)%ibdrf == 0)       ! a bidirectional surface: ISALB 7, 8, 9 (drt.f:468-470)
    cfg%ibdrf = recs(1)%ibdrf; cfg%bpar = recs(1)%bpar
    cfg%numu = numu; cfg%nphi = merge(view%nphi, 0, radcalc)
    cfg%nlevel_out = nlev; cfg%device = 0
    cfg%max_batch = max(1, npart)
    if (phase == 2) cfg%max_batch = max(256, npart)    ! (a batch's later runs of this configuration may be longer)
    cfg%corint = merge(1, 0, corr)
    cfg%umu0 = recs(1)%umu0; cfg%phi0 = phi0; cfg%fisot = fisot
    cfg%btemp = btemp; cfg%ttemp = ttemp; cfg%temis = temis
    cfg%temper = c_loc(temper); cfg%umu = c_loc(umu); cfg%phi = c_loc(phiv)
    cfg%level_out = c_loc(level_out)
    call system_clock(tick_c0)
    if (size(devices) == 0) then
      rc = sbd_fleet_create(cfg, 0, c_null_ptr, fl)
    else
      rc = sbd_fleet_create(cfg, int(size(devices), c_int32_t), c_loc(devices), fl)
    end if
    call system_clock(tick_c1)
    t_create = t_create + real(tick_c1 - tick_c0, 8)/real(tick_rate, 8)
    if (rc /= SBD_OK .and. rc /= SBD_E_RETRY_NSTR) &
      call quit('sbd_fleet_create: '//sbd_strerror_f(rc)//' '//sbd_last_error_f())
    if (aborted) then
      fl = c_null_ptr; rc = -1_c_int; return
    end if
    if (nslot < max_fleets) then
      nslot = nslot + 1
      k = nslot
    else                                               ! full: the fleet used longest ago makes room
      oldest = 1
      do k = 2, nslot
        if (slots(k)%age < slots(oldest)%age) oldest = k
      end do
      k = oldest
      if (c_associated(slots(k)%fleet)) call sbd_fleet_destroy(slots(k)%fleet)
    end if
    slots(k)%fleet = fl; slots(k)%rc = rc; slots(k)%max_batch = int(cfg%max_batch); slots(k)%age = clock
    slots(k)%key = key
  end function

  ! CHEKIN's report on a bidirectional surface whose flux albedo leaves [0,1] (disort.f:5080-5096: 101 incidence
  ! cosines, one line pair per offender, on stdout before the fatal message) for record k, then the fatal stop: the
  ! engine flags the item, the lines come from the same integral on the host (sbd_surface_flux_albedo)
  subroutine input_stop(k)
    use sbd_surface_mod, only: surface_model, flux_albedo
    integer, intent(in) :: k
    type(surface_model) :: sm
    integer :: irmu, lc, km, ipk, row
    real(kr) :: rmu, flxalb
    ! CHEKIN's reports on the layer arrays come first, in its order (disort.f:4947-4953: the albedo and its layer,
    ! then the variable's name, per offending layer; disort.f:4975-4982: one line per offending moment; the 50th
    ! line raises warning 12, disutil.f:345-346).  The band model itself can make such a record: a cloud table
    ! interpolated beyond its wavelengths returns a single-scattering albedo above one, and the reference stops there
    ! with the wavelengths before it on stdout.
    ipk = where_solved(k)
    nbad = 0
    if (ipk > 0 .and. use_mix) then                      ! the item's arguments as the device formed them
      allocate(one_dtau(nz), one_ssalb(nz), one_pmom(0:nmom, nz))
      if (gas_dev) then                                  ! (the item's gas depths: fetched from the devices after the solves)
        call assemble_item(mix, int(pmom_row(ipk)) + 1, gas_depths(:, kterm(ipk) + 1, int(pmom_row(ipk)) + 1), nmom, &
                           one_dtau, one_ssalb, one_pmom)
      else
        call assemble_item(mix, int(pmom_row(ipk)) + 1, dtauc(:, ipk), nmom, one_dtau, one_ssalb, one_pmom)
      end if
      do lc = 1, nz
        if (one_ssalb(lc) < 0._kr .or. one_ssalb(lc) > 1._kr) then
          print *, one_ssalb(lc), ', ', lc
          call write_bad('SSALB')
        end if
      end do
      do lc = 1, nz
        do km = 0, nmom
          if (one_pmom(km, lc) < -1._kr .or. one_pmom(km, lc) > 1._kr) call write_bad('PMOM')
        end do
      end do
    else if (ipk > 0) then
      do lc = 1, nz
        if (ssalb(lc, ipk) < 0._kr .or. ssalb(lc, ipk) > 1._kr) then
          print *, ssalb(lc, ipk), ', ', lc
          call write_bad('SSALB')
        end if
      end do
      row = merge(int(pmom_row(ipk)) + 1, ipk, from_model)
      do lc = 1, nz
        do km = 0, nmom
          if (pmom(km, lc, row) < -1._kr .or. pmom(km, lc, row) > 1._kr) call write_bad('PMOM')
        end do
      end do
    end if
    if (recs(k)%ibdrf == 0) then
      call warn_file(0, 'DISORT--input and/or dimension errors', phase == 0)
      return
    end if
    sm%ibdrf = recs(k)%ibdrf; sm%par = recs(k)%bpar
    do irmu = 0, 100
      rmu = real(irmu*0.01, kr)                         ! (IRMU*0.01 in default real, as the reference types it)
      flxalb = flux_albedo(sm, recs(k)%bitem, rmu)
      if (flxalb < 0._kr .or. flxalb > 1._kr) then
        call warn_file(8, 'DREF--albedo value not in (0,1)')
        print '(a,2es11.3)', 'mu, flxalb: ', rmu, flxalb
        call write_bad('FUNCTION BDREF')
      end if
    end do
    call warn_file(0, 'DISORT--input and/or dimension errors', phase == 0)
  end subroutine
  subroutine write_bad(name)
    character(len=*), intent(in) :: name
    write(*, '(3a)') ' ****  Input variable  ', name, '  in error  ****'
    nbad = nbad + 1
    if (nbad == 50) call warn_file(12, 'Too many input errors.  Aborting...')
  end subroutine

  subroutine release_fleets()
    call release_all_fleets()
  end subroutine

  ! gasset + depthscl for every spectral point of the run on the devices of `fl` (sbd_fleet_gas_terms): gas_nk, gas_wt,
  ! gas_fail; with_depths also brings the depths back (gas_depths(layer, k-term, point))
  subroutine gas_terms_on(fl, with_depths)
    type(c_ptr), intent(in) :: fl
    logical, intent(in) :: with_depths
    type(sbd_gas_model) :: gm
    integer(c_int) :: rcg
    integer :: np
    type(c_ptr) :: dptr, lptr
    type(sbd_scat_model) :: sm
    np = mix%npoint
    if (.not. allocated(gas_nk)) allocate(gas_nk(np), gas_fail(np), gas_wt(3, np))
    gm%nz = int(nz, c_int32_t); gm%kdist = int(mix%kdist, c_int32_t)
    gm%uu = c_loc(mix%uu); gm%z = c_loc(mix%z)
    gm%amu0_first = mix%amu_gas(1); gm%amu0_rest = mix%amu_gas(2); gm%xo4 = mix%xo4
    gm%tables = c_loc(tables_image); gm%tables_bytes = int(size(tables_image), c_size_t)
    dptr = c_null_ptr
    if (with_depths) then
      if (.not. allocated(gas_depths)) allocate(gas_depths(nz, 3, np))
      dptr = c_loc(gas_depths)
    end if
    if (scat_dev) then                                   ! the layer blocks are made on the devices too
      call scat_model_of(sm)
      lptr = c_null_ptr
      if (with_depths) then                              ! (CHEKIN's report on a refused item needs its block)
        deallocate(mix%lay)
        allocate(mix%lay(nz, mix%nch, np))
        lptr = c_loc(mix%lay)
      end if
      rcg = sbd_fleet_point_terms(fl, gm, sm, int(np, c_int32_t), c_loc(mix%wl), int(mix%nch, c_int32_t), &
                                  c_loc(gas_nk), c_loc(gas_wt), c_loc(gas_fail), dptr, lptr, gas_lay_token)
      if (rcg /= SBD_OK) call quit('sbd_fleet_point_terms: '//sbd_strerror_f(rcg)//' '//sbd_last_error_f())
      return
    end if
    rcg = sbd_fleet_gas_terms(fl, gm, int(np, c_int32_t), c_loc(mix%wl), c_loc(mix%lay), int(mix%nch, c_int32_t), &
                              c_loc(gas_nk), c_loc(gas_wt), c_loc(gas_fail), dptr, gas_lay_token)
    if (rcg /= SBD_OK) call quit('sbd_fleet_gas_terms: '//sbd_strerror_f(rcg)//' '//sbd_last_error_f())
  end subroutine

  ! the run's scatterers as the C ABI's model (pointers into `mix`, which outlives every call)
  subroutine scat_model_of(sm)
    type(sbd_scat_model), intent(out) :: sm
    sm%nz = int(nz, c_int32_t)
    sm%z = c_loc(mix%atm%z); sm%p = c_loc(mix%atm%p); sm%t = c_loc(mix%atm%t)
    sm%xrsc = mix%xrsc
    sm%cloud_term = int(mix%ncloud_term, c_int32_t)
    sm%cld_nslot = int(mix%deck%nslot, c_int32_t); sm%cld_layer = int(mix%deck%layer, c_int32_t)
    sm%cld_tcloud = mix%deck%tcloud; sm%cld_lwp = mix%deck%lwp; sm%cld_nre = mix%deck%nre
    sm%iaer = int(mix%load%iaer, c_int32_t); sm%nosct = int(mix%load%nosct, c_int32_t)
    sm%aer_nwl = int(mix%load%nwl, c_int32_t)
    sm%aer_wl = c_null_ptr; sm%aer_ext = c_null_ptr; sm%aer_absb = c_null_ptr; sm%aer_asym = c_null_ptr
    sm%aer_column = c_null_ptr
    if (mix%load%iaer /= 0) then
      sm%aer_wl = c_loc(mix%load%wl); sm%aer_ext = c_loc(mix%load%ext); sm%aer_absb = c_loc(mix%load%absb)
      sm%aer_asym = c_loc(mix%load%asym); sm%aer_column = c_loc(mix%load%column)
    end if
    sm%abaer = mix%load%abaer
    sm%nstrat = int(mix%load%nstrat, c_int32_t); sm%jaer = int(mix%load%jaer, c_int32_t)
    sm%strat_layer = int(mix%load%strat_layer, c_int32_t); sm%taerst = mix%load%taerst
    sm%tables = c_loc(tables_image); sm%tables_bytes = int(size(tables_image), c_size_t)
  end subroutine

  ! one point's scattering depths (channel 4 of its block) on the host, from the source the devices run
  function host_block_of(ipt) result(sc)
    integer, intent(in) :: ipt
    real(kr) :: sc(nz)
    real(kr), allocatable, target :: blk(:, :)
    real(kr), target :: w1(1)
    type(sbd_scat_model) :: sm
    integer(c_int) :: rcb
    allocate(blk(nz, mix%nch))
    call scat_model_of(sm)
    w1(1) = mix%wl(ipt)
    rcb = sbd_scatter_blocks_host(sm, 1_c_int32_t, c_loc(w1), int(mix%nch, c_int32_t), c_loc(blk))
    if (rcb /= SBD_OK) call quit('sbd_scatter_blocks_host: '//sbd_strerror_f(rcb))
    sc = blk(:, 4)
  end function

  ! solve batch positions p0..p1 on the run's GPUs; per-run formats also get their weighted sums
  subroutine solve_part(p0, p1, beam, corrections)
    integer, intent(in) :: p0, p1
    logical, intent(in) :: beam, corrections
    type(sbd_batch_in) :: bin
    type(sbd_mix_in) :: mxin
    type(sbd_batch_out) :: bout
    type(c_ptr) :: fleet, wptr, aptr, uptr
    integer(c_int) :: rc
    integer :: ntry, ns
    if (p1 < p0) return
    fleet = c_null_ptr
    do ntry = 0, 2
      ns = nstr + ntry*(3*ntry - 5)                    ! nstr, nstr-2, nstr+2
      if (ns < 4) cycle
      if (ns > nstrms) exit
      fleet = fleet_for(ns, corrections .and. radcalc .and. beam, rc)   ! (corrections off without a beam, disort.f:2695)
      if (aborted) return
      if (rc == SBD_OK) exit
      ! rc == SBD_E_RETRY_NSTR: SETDIS tests the beam angle only when there is a beam (disort.f:2641-2650)
      if (.not. beam) exit
      call warn_file(1, 'SETDIS--beam angle=computational angle; change NSTR')
      fleet = c_null_ptr
    end do
    if (.not. c_associated(fleet)) then
      write(*, *) 'Error --- NSTR dithering procedure failed'
      call leave(); return
    end if
    bout%flux = c_loc(flux(1, 1, p0)); bout%status = c_loc(status(p0))
    bout%uu = c_null_ptr
    if (radcalc) bout%uu = c_loc(uu(1, 1, 1, p0))
    wptr = c_null_ptr; aptr = c_null_ptr; uptr = c_null_ptr
    if (.not. fmt%per_point) then
      wptr = c_loc(weight(p0)); aptr = c_loc(acc_flux)
      if (radcalc) uptr = c_loc(acc_uu)
      ! (a per-run format reads the engine's sums, not the items: their fluxes and intensities stay on the device --
      !  16 MB per 200 000 items that would cross PCIe for nobody.  SBD_ORDERED_SUMS=1 adds the items on the host.)
      if (.not. items_wanted) then
        bout%flux = c_null_ptr; bout%uu = c_null_ptr
      end if
    end if
    if (use_mix) then
      ! compact form: the part's items with the gas of their k-term, the spectral points' blocks by their index in the
      ! run (the engine stages the blocks the part's items refer to); per point the band edges, the incident flux, the
      ! albedo and the thermal switch (the same for the k-terms of a point)
      mxin%nwork = p1 - p0 + 1; mxin%npoint = int(mix%npoint, c_int32_t)
      mxin%point_of = c_loc(pmom_row(p0))
      if (gas_dev) then                                  ! the depths are on the devices of gas_fleet: by (point, k-term)
        if (.not. c_associated(fleet, gas_fleet)) then   ! (another stream count after an NSTR retry: other engines)
          call gas_terms_on(fleet, .false.)
          if (aborted) return
          gas_fleet = fleet
        end if
        mxin%dtaug = c_null_ptr; mxin%kterm = c_loc(kterm(p0))
        mxin%lay_token = gas_lay_token                   ! the layer blocks that call left on the devices (mix%lay unchanged since)
      else
        mxin%dtaug = c_loc(dtauc(1, p0)); mxin%kterm = c_null_ptr
      end if
      mxin%nterm = mix%nterm; mxin%family = 0
      mxin%family(1:mix%nterm) = mix%family(1:mix%nterm)
      mxin%lay = c_null_ptr                              ! (made on the devices: lay_token names them)
      if (size(mix%lay, 3) > 0) mxin%lay = c_loc(mix%lay)
      mxin%wvnmlo = c_loc(pt_lo); mxin%wvnmhi = c_loc(pt_hi); mxin%fbeam = c_loc(pt_fb); mxin%albedo = c_loc(pt_al)
      mxin%plank = c_loc(pt_pl)
      rc = sbd_fleet_solve_mix_host(fleet, mxin, bout, wptr, aptr, uptr)
      if (rc /= SBD_OK) call quit('sbd_fleet_solve_mix_host: '//sbd_strerror_f(rc)//' '//sbd_last_error_f())
      return
    end if
    bin%nwork = p1 - p0 + 1
    bin%dtauc = c_loc(dtauc(1, p0)); bin%ssalb = c_loc(ssalb(1, p0))
    bin%wvnmlo = c_loc(wvnmlo(p0)); bin%wvnmhi = c_loc(wvnmhi(p0)); bin%fbeam = c_loc(fbeam(p0))
    bin%albedo = c_loc(albedo(p0)); bin%plank = c_loc(plank(p0))
    bin%bitem = c_null_ptr
    if (recs(1)%ibdrf == 1) bin%bitem = c_loc(bitem(1, p0))
    if (from_model) then                               ! moments per wavelength, shared by the k-terms
      bin%pmom = c_loc(pmom(0, 1, 1)); bin%pmom_row = c_loc(pmom_row(p0)); bin%npmom = int(size(pmom, 3), c_int32_t)
    else                                               ! (per item: only here is p0 a valid third index of pmom)
      bin%pmom = c_loc(pmom(0, 1, p0))
    end if
    rc = sbd_fleet_solve_host(fleet, bin, bout, wptr, aptr, uptr)
    if (rc /= SBD_OK) call quit('sbd_fleet_solve_host: '//sbd_strerror_f(rc)//' '//sbd_last_error_f())
    if (aborted) return
  end subroutine
end subroutine run_once

integer function nslot_now()
  use sbd_fleet_cache_mod, only: nslot
  nslot_now = nslot
end function

subroutine release_all_fleets()
  use iso_c_binding
  use sbd_engine_mod
  use sbd_fleet_cache_mod
  integer :: k
  do k = 1, nslot
    if (c_associated(slots(k)%fleet)) call sbd_fleet_destroy(slots(k)%fleet)
    slots(k)%fleet = c_null_ptr
    if (allocated(slots(k)%key)) deallocate(slots(k)%key)
  end do
  nslot = 0
end subroutine

! sbdart_amd --batch LIST: every directory of LIST (one per line, each with its INPUT) is one run; its text goes to
! SBDART.stdout in that directory, warning files beside it as always.  A pool of worker processes (forked before this
! process touches the GPU) runs phase 1 of the runs, a child per run -- the reference's own process-per-run isolation,
! STOPs included, at the price of a fork; this process owns the GPU and runs phase 2 of the runs in LIST order as their
! work items arrive.
subroutine run_batch(listfile)
  use iso_c_binding
  use sbd_tables_mod, only: tables_load
  use omp_lib
  character(len=*), intent(in) :: listfile
  interface
    integer(c_int) function sbd_px_chdir(path) bind(C, name='sbd_px_chdir')
      import; character(kind=c_char), intent(in) :: path(*)
    end function
    integer(c_int) function sbd_px_stdout_to(path, append) bind(C, name='sbd_px_stdout_to')
      import; character(kind=c_char), intent(in) :: path(*); integer(c_int), value :: append
    end function
    integer(c_int) function sbd_px_stderr_to(path) bind(C, name='sbd_px_stderr_to')
      import; character(kind=c_char), intent(in) :: path(*)
    end function
    integer(c_int) function sbd_px_fork() bind(C, name='sbd_px_fork')
      import
    end function
    integer(c_int) function sbd_px_wait(pid) bind(C, name='sbd_px_wait')
      import; integer(c_int), value :: pid
    end function
    subroutine sbd_px_exit_now(code) bind(C, name='sbd_px_exit_now')
      import; integer(c_int), value :: code
    end subroutine
    integer(c_int) function sbd_px_exists(path) bind(C, name='sbd_px_exists')
      import; character(kind=c_char), intent(in) :: path(*)
    end function
    integer(c_int) function sbd_px_remove(path) bind(C, name='sbd_px_remove')
      import; character(kind=c_char), intent(in) :: path(*)
    end function
    integer(c_int) function sbd_px_touch(path, value) bind(C, name='sbd_px_touch')
      import; character(kind=c_char), intent(in) :: path(*); integer(c_int), value :: value
    end function
    integer(c_int) function sbd_px_rename(from, to) bind(C, name='sbd_px_rename')
      import; character(kind=c_char), intent(in) :: from(*), to(*)
    end function
    integer(c_int) function sbd_px_append_line(path, text) bind(C, name='sbd_px_append_line')
      import; character(kind=c_char), intent(in) :: path(*), text(*)
    end function
    integer(c_int) function sbd_px_stdout_save() bind(C, name='sbd_px_stdout_save')
      import
    end function
    integer(c_int) function sbd_px_stdout_restore(fd) bind(C, name='sbd_px_stdout_restore')
      import; integer(c_int), value :: fd
    end function
    subroutine sbd_px_usleep(us) bind(C, name='sbd_px_usleep')
      import; integer(c_int), value :: us
    end subroutine
    integer(c_int) function sbd_px_ncpu() bind(C, name='sbd_px_ncpu')
      import
    end function
    integer(c_int) function sbd_px_getcwd(buf, n) bind(C, name='sbd_px_getcwd')
      import; character(kind=c_char) :: buf(*); integer(c_int), value :: n
    end function
  end interface
  character(len=1024), allocatable :: dirs(:)
  character(len=1024) :: line, home
  character(len=16) :: txt
  integer :: u, ios, nrun, i, k, nw, tlen, tstat
  integer(kind=8) :: c0, c1, crate
  integer(c_int) :: pid, rc, fd1
  integer(c_int), allocatable :: wpid(:)
  logical :: ok, phase1_only
  character(len=256) :: why

  call get_environment_variable('SBD_BATCH_PHASE1_ONLY', txt, tlen, tstat)
  phase1_only = tstat == 0 .and. tlen > 0
  k = sbd_px_getcwd(home, int(len(home), c_int))
  if (k < 0) stop 'sbdart_amd --batch: getcwd failed'
  home = home(1:k)
  open(newunit=u, file=listfile, status='old', iostat=ios)
  if (ios /= 0) stop 'sbdart_amd --batch: cannot open the list of run directories'
  nrun = 0
  do
    read(u, '(a)', iostat=ios) line
    if (ios /= 0) exit
    if (len_trim(line) > 0) nrun = nrun + 1
  end do
  rewind(u)
  allocate(dirs(nrun))
  i = 0
  do
    read(u, '(a)', iostat=ios) line
    if (ios /= 0) exit
    if (len_trim(line) == 0) cycle
    i = i + 1
    line = adjustl(line)
    if (line(1:1) == '/') then
      dirs(i) = line
    else
      dirs(i) = trim(home)//'/'//trim(line)
    end if
  end do
  close(u)
  do i = 1, nrun                                        ! (markers of an earlier batch in the same directories)
    call forget(i)
  end do
  call tables_load(ok, why)                             ! once, before the fork: the children inherit the tables
  nw = min(max(1, int(sbd_px_ncpu()) - 1), 16, max(1, nrun))
  call get_environment_variable('SBD_BATCH_WORKERS', txt, tlen, tstat)
  if (tstat == 0 .and. tlen > 0) read(txt(1:tlen), *, iostat=ios) nw
  nw = max(1, min(nw, max(1, nrun)))
  allocate(wpid(nw))
  flush(6)
  do k = 1, nw
    pid = sbd_px_fork()
    if (pid < 0) stop 'sbdart_amd --batch: fork failed'
    if (pid == 0) then                                  ! worker k: runs k, k+nw, ... -- a child per run
      call omp_set_num_threads(1)                       ! (the pool is the parallelism; a child's band model runs on one core)
      do i = k, nrun, nw
        call first_phase(i)
      end do
      call sbd_px_exit_now(0_c_int)
    end if
    wpid(k) = pid
  end do
  ! this process: the GPU side, runs in order (file descriptor 1 is re-pointed per run: the caller's is kept and put back)
  flush(6)
  fd1 = sbd_px_stdout_save()
  do i = 1, nrun
    call second_phase(i)
  end do
  flush(6)
  if (fd1 >= 0) rc = sbd_px_stdout_restore(fd1)
  do k = 1, nw
    rc = sbd_px_wait(wpid(k))
  end do
  call get_environment_variable('SBD_TIMING', txt, tlen, tstat)
  if (tstat == 0 .and. tlen > 0) write(0, '(a,i0,a,f8.3,a,f8.3,a,f8.3,a,i0,a)') 'sbdart_amd --batch: ', nrun, &
    ' runs; waiting for phase 1 ', t_wait, ' s; phase 2 ', t_phase2, ' s of which engine calls ', t_engine, ' s; ', &
    nslot_now(), ' fleets kept'
  call release_all_fleets()
  k = sbd_px_chdir(trim(home)//c_null_char)

contains

  ! (one procedure call per run: the path strings built here are stack temporaries (-fstack-arrays) that live until the
  !  procedure returns -- in a loop of the caller they piled up to a stack overflow after ~2 000 runs)
  subroutine forget(irun)
    integer, intent(in) :: irun
    integer :: r
    r = sbd_px_remove(trim(dirs(irun))//'/'//phase1_mark//c_null_char)
    r = sbd_px_remove(trim(dirs(irun))//'/'//items_file//c_null_char)
    r = sbd_px_remove(trim(dirs(irun))//'/'//items_tmp//c_null_char)
  end subroutine

  subroutine first_phase(irun)
    integer, intent(in) :: irun
    integer(c_int) :: child, code
    child = sbd_px_fork()
    if (child == 0) then
      if (sbd_px_chdir(trim(dirs(irun))//c_null_char) /= 0) call sbd_px_exit_now(3_c_int)
      if (sbd_px_stdout_to(stdout_file//c_null_char, 0_c_int) /= 0) call sbd_px_exit_now(3_c_int)
      if (sbd_px_stderr_to(stderr_file//c_null_char) /= 0) call sbd_px_exit_now(3_c_int)
      call run_once(1)
      flush(6)
      call sbd_px_exit_now(0_c_int)
    end if
    code = -1
    if (child > 0) code = sbd_px_wait(child)
    ! The work items become visible to phase 2 only COMPLETE: the child wrote them under a temporary name, and they get
    ! their name here once the child has exited with code 0.  A child that was killed or stopped on the way (a STOP of
    ! the model code after the file was opened, a signal) leaves no items: phase 2 skips the run, what the child printed
    ! stays in SBDART.stdout / SBDART.stderr like the reference's own process would have left it.
    if (sbd_px_exists(trim(dirs(irun))//'/'//items_tmp//c_null_char) /= 0) then
      if (code == 0) then
        if (sbd_px_rename(trim(dirs(irun))//'/'//items_tmp//c_null_char, trim(dirs(irun))//'/'//items_file//c_null_char) /= 0) code = -2
      else
        if (sbd_px_remove(trim(dirs(irun))//'/'//items_tmp//c_null_char) /= 0) continue
      end if
    end if
    if (code >= 128 .or. code < 0) then
      write(txt, '(i0)') code
      if (sbd_px_append_line(trim(dirs(irun))//'/'//stderr_file//c_null_char, &
          'sbdart_amd --batch: the first phase of this run ended with code '//trim(txt)//'; run skipped'//c_null_char) /= 0) continue
    end if
    if (sbd_px_touch(trim(dirs(irun))//'/'//phase1_mark//c_null_char, code) /= 0) continue
  end subroutine

  subroutine second_phase(irun)
    integer, intent(in) :: irun
    character(len=:), allocatable :: mark, items
    integer :: polls, r
    mark = trim(dirs(irun))//'/'//phase1_mark//c_null_char
    items = trim(dirs(irun))//'/'//items_file//c_null_char
    polls = 0
    call system_clock(c0, crate)
    do while (sbd_px_exists(mark) == 0)
      call sbd_px_usleep(100_c_int)
      polls = polls + 1
      if (polls > 6000000) stop 'sbdart_amd --batch: a run never finished its first phase'
    end do
    call system_clock(c1)
    t_wait = t_wait + real(c1 - c0, 8)/real(crate, 8)
    if (.not. phase1_only) then                         ! (tests without a GPU: the work items stay where phase 1 left them)
      if (sbd_px_exists(items) /= 0) then
        if (sbd_px_chdir(trim(dirs(irun))//c_null_char) /= 0) stop 'sbdart_amd --batch: cannot enter a run directory'
        flush(6)
        if (sbd_px_stdout_to(stdout_file//c_null_char, 1_c_int) /= 0) stop 'sbdart_amd --batch: cannot append to SBDART.stdout'
        call system_clock(c0)
        call run_once(2)
        flush(6)
        call system_clock(c1)
        t_phase2 = t_phase2 + real(c1 - c0, 8)/real(crate, 8)
        r = sbd_px_remove(items_file//c_null_char)
        r = sbd_px_remove(items_file//'.atm'//c_null_char)
      end if
    end if
    r = sbd_px_remove(mark)
  end subroutine
end subroutine

! sbdart_amd --serve SOCKET: a resident process that owns the GPU, the HIP runtime and the engines of the configurations it
! has met, serving `sbdart` clients (sbdart_client.c) one run at a time: the harnesses launch `sbdart` once per run in the
! run's directory and read its stdout (RunRT.py:2021-2044, TestRuns/test_runs:31-145) -- unchanged.  A client hands over its
! working directory and its own file descriptors 1 and 2; the run writes straight into them.
!   * the run's first phase (INPUT -> screening -> band model -> work items) in a CHILD, forked by a helper process that was
!     itself forked before this process touched the GPU: the reference's process-per-run isolation, STOPs of the model code
!     included, for a fork (as `--batch`);
!   * its second phase (work items -> engine -> text) here, on the engines kept from earlier runs;
!   * a run too large for a file of work items (the child says so before its band model: exit code 77) is made whole here
!     (phase 3: compact form, gas terms on the device); a run of middle size (more than SBDART_AMD_MID_MB = 1 MB of work
!     items) is REHEARSED in the child -- screening and band model without a STOP, exit code 78 -- and then made whole here
!     as well: no file of work items is written and read back;
!   * after SBDART_AMD_IDLE_S seconds (default 300) without a client the server leaves and removes its socket.
subroutine run_server(sockpath)
  use iso_c_binding
  use sbd_tables_mod, only: tables_load
  use omp_lib
  character(len=*), intent(in) :: sockpath
  interface
    integer(c_int) function sbd_px_chdir(path) bind(C, name='sbd_px_chdir')
      import; character(kind=c_char), intent(in) :: path(*)
    end function
    integer(c_int) function sbd_px_fork() bind(C, name='sbd_px_fork')
      import
    end function
    integer(c_int) function sbd_px_wait(pid) bind(C, name='sbd_px_wait')
      import; integer(c_int), value :: pid
    end function
    subroutine sbd_px_exit_now(code) bind(C, name='sbd_px_exit_now')
      import; integer(c_int), value :: code
    end subroutine
    integer(c_int) function sbd_px_exists(path) bind(C, name='sbd_px_exists')
      import; character(kind=c_char), intent(in) :: path(*)
    end function
    integer(c_int) function sbd_px_remove(path) bind(C, name='sbd_px_remove')
      import; character(kind=c_char), intent(in) :: path(*)
    end function
    integer(c_int) function sbd_px_rename(from, to) bind(C, name='sbd_px_rename')
      import; character(kind=c_char), intent(in) :: from(*), to(*)
    end function
    integer(c_int) function sbd_px_dup(fd) bind(C, name='sbd_px_dup')
      import; integer(c_int), value :: fd
    end function
    integer(c_int) function sbd_px_dup2(from, to) bind(C, name='sbd_px_dup2')
      import; integer(c_int), value :: from, to
    end function
    integer(c_int) function sbd_px_close(fd) bind(C, name='sbd_px_close')
      import; integer(c_int), value :: fd
    end function
    integer(c_int) function sbd_sv_socketpair(sv) bind(C, name='sbd_sv_socketpair')
      import; integer(c_int) :: sv(2)
    end function
    integer(c_int) function sbd_sv_listen(path) bind(C, name='sbd_sv_listen')
      import; character(kind=c_char), intent(in) :: path(*)
    end function
    integer(c_int) function sbd_sv_accept(lfd, idle_ms, dir, dirlen, fd1, fd2) bind(C, name='sbd_sv_accept')
      import; integer(c_int), value :: lfd, idle_ms, dirlen; character(kind=c_char) :: dir(*); integer(c_int) :: fd1, fd2
    end function
    integer(c_int) function sbd_sv_send_job(sock, dir, fd1, fd2) bind(C, name='sbd_sv_send_job')
      import; integer(c_int), value :: sock, fd1, fd2; character(kind=c_char), intent(in) :: dir(*)
    end function
    integer(c_int) function sbd_sv_recv_job(sock, dir, dirlen, fd1, fd2) bind(C, name='sbd_sv_recv_job')
      import; integer(c_int), value :: sock, dirlen; character(kind=c_char) :: dir(*); integer(c_int) :: fd1, fd2
    end function
    integer(c_int) function sbd_sv_send_code(sock, code) bind(C, name='sbd_sv_send_code')
      import; integer(c_int), value :: sock, code
    end function
    integer(c_int) function sbd_sv_recv_code(sock, code) bind(C, name='sbd_sv_recv_code')
      import; integer(c_int), value :: sock; integer(c_int) :: code
    end function
  end interface
  character(kind=c_char, len=1024), target :: dir
  character(len=16) :: txt
  integer :: tlen, tstat, ios, idle_s, nserved, dlen
  integer(c_int) :: sv(2), helper, lfd, cfd, fd1, fd2, code, rc, keep1, keep2
  integer(kind=8) :: c0, c1, crate
  real(kind=8) :: t_first, t_second
  logical :: ok, timing
  character(len=256) :: why

  serve_mode = .true.
  idle_s = 300
  call get_environment_variable('SBDART_AMD_IDLE_S', txt, tlen, tstat)
  if (tstat == 0 .and. tlen > 0) read(txt(1:tlen), *, iostat=ios) idle_s
  call get_environment_variable('SBDART_AMD_BIG_MB', txt, tlen, tstat)
  if (tstat == 0 .and. tlen > 0) then
    read(txt(1:tlen), *, iostat=ios) serve_big
    serve_big = serve_big*1.0d6
  end if
  call get_environment_variable('SBDART_AMD_MID_MB', txt, tlen, tstat)
  if (tstat == 0 .and. tlen > 0) then
    read(txt(1:tlen), *, iostat=ios) serve_mid
    serve_mid = serve_mid*1.0d6
  end if
  call get_environment_variable('SBD_TIMING', txt, tlen, tstat)
  timing = tstat == 0 .and. tlen > 0
  call tables_load(ok, why)                             ! once: the helper and its children inherit the tables
  if (sbd_sv_socketpair(sv) /= 0) stop 'sbdart_amd --serve: socketpair failed'
  flush(6)
  helper = sbd_px_fork()                                ! BEFORE this process touches the GPU (a HIP context does not survive a fork)
  if (helper < 0) stop 'sbdart_amd --serve: fork failed'
  if (helper == 0) then
    rc = sbd_px_close(sv(1))
    call helper_loop(sv(2))
    call sbd_px_exit_now(0_c_int)
  end if
  rc = sbd_px_close(sv(2))
  lfd = sbd_sv_listen(trim(sockpath)//c_null_char)
  if (lfd < 0) then
    rc = sbd_px_close(sv(1))
    rc = sbd_px_wait(helper)
    if (lfd == -2) then
      write(0, '(a)') 'sbdart_amd --serve: another server is listening on this socket'
      call sbd_px_exit_now(4_c_int)
    end if
    write(0, '(a)') 'sbdart_amd --serve: cannot listen on the socket path'
    call sbd_px_exit_now(5_c_int)
  end if
  nserved = 0; t_first = 0; t_second = 0
  call system_clock(c0, crate)
  do
    cfd = sbd_sv_accept(lfd, int(min(idle_s, 2000000)*1000, c_int), dir, int(len(dir), c_int), fd1, fd2)
    if (cfd < 0) exit                                   ! idle (or the socket broke): leave
    dlen = index(dir, c_null_char) - 1
    call serve_one(dir(1:dlen))
    rc = sbd_px_close(fd1); rc = sbd_px_close(fd2)
    rc = sbd_sv_send_code(cfd, code)
    rc = sbd_px_close(cfd)
    nserved = nserved + 1
  end do
  rc = sbd_px_close(lfd)
  rc = sbd_px_remove(trim(sockpath)//c_null_char)
  rc = sbd_px_close(sv(1))                              ! (the helper sees the end of its channel and leaves)
  rc = sbd_px_wait(helper)
  if (timing) write(0, '(a,i0,a,f9.3,a,f9.3,a,f9.3,a,i0,a)') 'sbdart_amd --serve: ', nserved, ' runs; first phases ', t_first, &
    ' s, second phases ', t_second, ' s of which engine calls ', t_engine, ' s; ', nslot_now(), ' fleets kept'
  call release_all_fleets()

contains

  ! the fork helper: a child per run, in the run's directory, writing into the client's descriptors
  subroutine helper_loop(sock)
    integer(c_int), intent(in) :: sock
    character(kind=c_char, len=1024), target :: d
    integer(c_int) :: a, b, child, cd, r
    integer :: n
    do
      if (sbd_sv_recv_job(sock, d, int(len(d), c_int), a, b) /= 0) exit
      n = index(d, c_null_char) - 1
      child = sbd_px_fork()
      if (child == 0) then
        r = sbd_px_close(sock)
        if (sbd_px_chdir(d(1:n)//c_null_char) /= 0) call sbd_px_exit_now(3_c_int)
        if (sbd_px_dup2(a, 1_c_int) /= 0 .or. sbd_px_dup2(b, 2_c_int) /= 0) call sbd_px_exit_now(3_c_int)
        r = sbd_px_close(a); r = sbd_px_close(b)
        call omp_set_num_threads(min(8, max(1, omp_get_num_procs())))
        call run_once(1)
        flush(6)
        call sbd_px_exit_now(0_c_int)
      end if
      r = sbd_px_close(a); r = sbd_px_close(b)
      cd = -1
      if (child > 0) cd = sbd_px_wait(child)
      ! (the work items become visible only COMPLETE, as in a batch: the child wrote them under a temporary name)
      if (sbd_px_exists(d(1:n)//'/'//items_tmp//c_null_char) /= 0) then
        if (cd == 0) then
          if (sbd_px_rename(d(1:n)//'/'//items_tmp//c_null_char, d(1:n)//'/'//items_file//c_null_char) /= 0) cd = -2
        else
          r = sbd_px_remove(d(1:n)//'/'//items_tmp//c_null_char)
        end if
      end if
      if (sbd_sv_send_code(sock, cd) /= 0) exit
    end do
  end subroutine

  ! one client's run; sets `code`, the client's exit code
  subroutine serve_one(d)
    character(len=*), intent(in) :: d
    integer(c_int) :: r, c1st
    integer(kind=8) :: t0, t1, t2
    r = sbd_px_remove(d//'/'//items_file//c_null_char)      ! (left-overs of a run that was killed)
    r = sbd_px_remove(d//'/'//items_tmp//c_null_char)
    call system_clock(t0)
    code = 1
    if (sbd_sv_send_job(sv(1), d//c_null_char, fd1, fd2) /= 0) return
    if (sbd_sv_recv_code(sv(1), c1st) /= 0) return
    call system_clock(t1)
    t_first = t_first + real(t1 - t0, 8)/real(crate, 8)
    code = c1st
    if (c1st == 77 .or. c1st == 78 .or. (c1st == 0 .and. sbd_px_exists(d//'/'//items_file//c_null_char) /= 0)) then
      if (sbd_px_chdir(d//c_null_char) /= 0) then
        code = 3; return
      end if
      flush(6); flush(0)
      keep1 = sbd_px_dup(1_c_int); keep2 = sbd_px_dup(2_c_int)
      if (sbd_px_dup2(fd1, 1_c_int) == 0 .and. sbd_px_dup2(fd2, 2_c_int) == 0) then
        call run_once(merge(3, 2, c1st == 77 .or. c1st == 78))
        flush(6); flush(0)
        code = 0
      else
        code = 3
      end if
      r = sbd_px_dup2(keep1, 1_c_int); r = sbd_px_dup2(keep2, 2_c_int)
      r = sbd_px_close(keep1); r = sbd_px_close(keep2)
      r = sbd_px_remove(items_file//c_null_char)
      r = sbd_px_remove(items_file//'.atm'//c_null_char)
      r = sbd_px_chdir('/'//c_null_char)
    end if
    call system_clock(t2)
    t_second = t_second + real(t2 - t1, 8)/real(crate, 8)
  end subroutine
end subroutine

end module sbd_run_mod

program sbdart_amd
  use iso_c_binding
  use sbd_run_mod
  implicit none
  character(len=1024) :: arg, list
  integer :: n
  interface
    integer(c_int) function sbd_px_setenv_default(name, value) bind(C, name='sbd_px_setenv_default')
      use iso_c_binding
      character(kind=c_char), intent(in) :: name(*), value(*)
    end function
    integer(c_int) function sbd_px_restrict_devices() bind(C, name='sbd_px_restrict_devices')
      use iso_c_binding
    end function
  end interface
  call system_clock(tick_program)
  ! The OpenMP runtime's first act is to map the machine's topology for thread affinity: 35-70 ms on the 256-core GPU
  ! box (profiles/r05_e2e_omp_init.txt) -- ten times the band model's wavelength loop, which needs no placement.
  ! Unless the user says otherwise, affinity is off.
  n = sbd_px_setenv_default('KMP_AFFINITY'//c_null_char, 'disabled'//c_null_char)
  ! The HIP runtime brings up every GPU it can see at its first call: the run's devices only (sbd_posix.c)
  n = sbd_px_restrict_devices()
  n = command_argument_count()
  if (n >= 2) then
    call get_command_argument(1, arg)
    if (trim(arg) == '--batch') then
      call get_command_argument(2, list)
      call run_batch(trim(list))
      stop
    end if
    if (trim(arg) == '--serve') then
      call get_command_argument(2, list)
      call run_server(trim(list))
      stop
    end if
  end if
  if (n >= 1) then
    write(0, '(a)') 'usage: sbdart_amd            (one run: ./INPUT -> stdout, as the reference)'
    write(0, '(a)') '       sbdart_amd --batch LIST   (LIST: run directories, one per line; text -> <dir>/SBDART.stdout)'
    write(0, '(a)') '       sbdart_amd --serve SOCKET (resident server of the `sbdart` client beside this file: RunRT / TestRuns unchanged)'
    stop 2
  end if
  call run_once(0)
end program sbdart_amd
