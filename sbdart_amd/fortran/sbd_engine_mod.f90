! ISO_C_BINDING view of include/sbdart_amd.h -- the thin shim through which the
! Fortran-2003 host (program sbdart_amd, or the reference's own drt.f, see
! INTEGRATION.md) drives the MI355X engine.  One type per C struct, one interface
! per C entry point; no logic lives here.
module sbd_engine_mod
  use iso_c_binding
  implicit none
  private
  public :: sbd_run_cfg, sbd_batch_in, sbd_batch_out, sbd_mix_in, sbd_gas_model, sbd_fleet_gas_terms, &
            sbd_scat_model, sbd_fleet_point_terms, sbd_scatter_blocks_host
  public :: sbd_engine_create, sbd_engine_destroy, sbd_engine_solve_host, &
            sbd_engine_solve_device, sbd_engine_accumulate_host, sbd_engine_nlevel, &
            sbd_engine_chunk, sbd_strerror_f, sbd_last_error_f, sbd_abi_version
  public :: sbd_fleet_create, sbd_fleet_destroy, sbd_fleet_size, sbd_fleet_uses_rccl, sbd_shard_range, &
            sbd_fleet_solve_host, sbd_fleet_solve_mix_host
  public :: SBD_OK, SBD_E_RETRY_NSTR, SBD_NFLUX, SBD_ABI_VER, SBD_MIX_MAX_TERMS
  public :: SBD_ST_WARN_SOLVE0, SBD_ST_WARN_UPBEAM, SBD_ST_WARN_UPISOT, SBD_ST_ERR_EIGEN, &
            SBD_ST_RETRY_NSTR, SBD_ST_ERR_INPUT, SBD_ST_WARN_PLKAVG, SBD_ST_WARN_PLKCONV

  integer(c_int), parameter :: SBD_ABI_VER = 7, SBD_OK = 0, SBD_E_RETRY_NSTR = -2, SBD_NFLUX = 5
  integer(c_int), parameter :: SBD_ST_WARN_SOLVE0 = 1, SBD_ST_WARN_UPBEAM = 2, SBD_ST_WARN_UPISOT = 4, &
       SBD_ST_ERR_EIGEN = 8, SBD_ST_RETRY_NSTR = 16, SBD_ST_ERR_INPUT = 32, SBD_ST_WARN_PLKAVG = 64, SBD_ST_WARN_PLKCONV = 128

  type, bind(C) :: sbd_run_cfg
    integer(c_int32_t) :: abi_version, nlyr, nstr, nmom, onlyfl, lamber, usrang, numu, nphi, &
                          nlevel_out, device, max_batch, corint = 0, ibdrf = 0
    real(c_double) :: umu0, phi0, fisot, btemp, ttemp, temis
    type(c_ptr) :: temper, umu, phi, level_out
    real(c_double) :: bpar(8) = 0        ! bidirectional surface parameters (lamber = 0)
    integer(c_int32_t) :: ibcnd = 0, pivot_exact = 0   ! (pivot_exact = 1: LINPACK's exact pivot rule for NSTR <= 16) 1: albedo / transmissivity of the medium (ALBTRN); SBDART never sets it
  end type

  type, bind(C) :: sbd_batch_in
    integer(c_int32_t) :: nwork
    type(c_ptr) :: dtauc, ssalb, pmom, wvnmlo, wvnmhi, fbeam, albedo, plank
    type(c_ptr) :: bitem = c_null_ptr    ! [4, nwork] ocean surface constants, else null
    type(c_ptr) :: pmom_row = c_null_ptr ! [nwork] 0-based block of pmom per item when the k-terms share their moments
    integer(c_int32_t) :: npmom = 0      ! blocks in pmom (pmom_row given)
  end type

  type, bind(C) :: sbd_batch_out
    type(c_ptr) :: flux, uu, status
    type(c_ptr) :: albtrn = c_null_ptr   ! ibcnd = 1 only
  end type

  ! a batch in compact form (include/sbdart_amd.h, ABI v6): per spectral point a block lay(nlyr, 4 + 3 nterm) -- cloud,
  ! aerosol and Rayleigh depths, their scattering depth, then g, m1, m2 of every scattering term -- and per work item
  ! the gas of its k-term; the engine assembles DTAUC / SSALB / PMOM on the device in the reference's own association
  integer, parameter :: SBD_MIX_MAX_TERMS = 6
  type, bind(C) :: sbd_mix_in
    integer(c_int32_t) :: nwork, npoint
    type(c_ptr) :: point_of, dtaug
    integer(c_int32_t) :: nterm = 0
    integer(c_int32_t) :: family(SBD_MIX_MAX_TERMS) = 0
    type(c_ptr) :: lay, wvnmlo, wvnmhi, fbeam, albedo, plank
    type(c_ptr) :: kterm = c_null_ptr    ! with dtaug = c_null_ptr: the items' k-terms (0-based); their gas depths are
                                         ! the ones sbd_fleet_gas_terms left on the devices
    integer(c_int64_t) :: lay_token = 0  ! (ABI v7) the number sbd_fleet_gas_terms returned for the layer blocks it left
                                         ! on the devices: the solve reads those; 0 = stage them from `lay`
  end type

  ! the gas part of the band model for a run (include/sbdart_amd.h): evaluated by the engine for all wavelengths at once
  type, bind(C) :: sbd_gas_model
    integer(c_int32_t) :: nz, kdist
    type(c_ptr) :: uu, z                 ! (63, nz) absorber amounts, (nz) altitudes, levels bottom-up
    real(c_double) :: amu0_first, amu0_rest, xo4
    type(c_ptr) :: tables                ! image of sbdart_tables.bin
    integer(c_size_t) :: tables_bytes
  end type

  ! the scatterers' part of the band model for a run (include/sbdart_amd.h): the layer blocks made on the devices
  type, bind(C) :: sbd_scat_model
    integer(c_int32_t) :: nz
    type(c_ptr) :: z, p, t               ! (nz) levels bottom-up
    real(c_double) :: xrsc
    integer(c_int32_t) :: cloud_term, cld_nslot, cld_layer(5)
    real(c_double) :: cld_tcloud(5), cld_lwp(5), cld_nre(5)
    integer(c_int32_t) :: iaer, nosct, aer_nwl
    type(c_ptr) :: aer_wl, aer_ext, aer_absb, aer_asym
    real(c_double) :: abaer
    type(c_ptr) :: aer_column            ! (nz) layers top-down
    integer(c_int32_t) :: nstrat, jaer(5), strat_layer(5)
    real(c_double) :: taerst(5)
    type(c_ptr) :: tables
    integer(c_size_t) :: tables_bytes
  end type

  interface
    function sbd_engine_create(cfg, eng) bind(C, name='sbd_engine_create') result(rc)
      import
      type(sbd_run_cfg), intent(in) :: cfg
      type(c_ptr), intent(out) :: eng
      integer(c_int) :: rc
    end function
    subroutine sbd_engine_destroy(eng) bind(C, name='sbd_engine_destroy')
      import
      type(c_ptr), value :: eng
    end subroutine
    function sbd_engine_solve_host(eng, bin, bout) bind(C, name='sbd_engine_solve_host') result(rc)
      import
      type(c_ptr), value :: eng
      type(sbd_batch_in), intent(in) :: bin
      type(sbd_batch_out), intent(in) :: bout
      integer(c_int) :: rc
    end function
    function sbd_engine_solve_device(eng, bin, bout, stream) bind(C, name='sbd_engine_solve_device') result(rc)
      import
      type(c_ptr), value :: eng, stream
      type(sbd_batch_in), intent(in) :: bin
      type(sbd_batch_out), intent(in) :: bout
      integer(c_int) :: rc
    end function
    function sbd_engine_accumulate_host(eng, nwork, weight, flux, uu, acc_flux, acc_uu) &
         bind(C, name='sbd_engine_accumulate_host') result(rc)
      import
      type(c_ptr), value :: eng, weight, flux, uu, acc_flux, acc_uu
      integer(c_int32_t), value :: nwork
      integer(c_int) :: rc
    end function
    ! ---- several GPUs from one process: one engine per device, one reduce of the sums ----
    function sbd_fleet_create(cfg, ndev, devices, fleet) bind(C, name='sbd_fleet_create') result(rc)
      import
      type(sbd_run_cfg), intent(in) :: cfg
      integer(c_int32_t), value :: ndev
      type(c_ptr), value :: devices            ! int32 device ordinals, or c_null_ptr: every visible device
      type(c_ptr), intent(out) :: fleet
      integer(c_int) :: rc
    end function
    subroutine sbd_fleet_destroy(fleet) bind(C, name='sbd_fleet_destroy')
      import
      type(c_ptr), value :: fleet
    end subroutine
    function sbd_fleet_size(fleet) bind(C, name='sbd_fleet_size') result(n)
      import
      type(c_ptr), value :: fleet
      integer(c_int32_t) :: n
    end function
    function sbd_fleet_uses_rccl(fleet) bind(C, name='sbd_fleet_uses_rccl') result(n)
      import
      type(c_ptr), value :: fleet
      integer(c_int32_t) :: n
    end function
    subroutine sbd_shard_range(nwork, nshard, rank, lo, hi) bind(C, name='sbd_shard_range')
      import
      integer(c_int32_t), value :: nwork, nshard, rank
      integer(c_int32_t), intent(out) :: lo, hi
    end subroutine
    function sbd_fleet_solve_host(fleet, bin, bout, weight, acc_flux, acc_uu) &
         bind(C, name='sbd_fleet_solve_host') result(rc)
      import
      type(c_ptr), value :: fleet, weight, acc_flux, acc_uu
      type(sbd_batch_in), intent(in) :: bin
      type(sbd_batch_out), intent(in) :: bout
      integer(c_int) :: rc
    end function
    function sbd_fleet_solve_mix_host(fleet, min, bout, weight, acc_flux, acc_uu) &
         bind(C, name='sbd_fleet_solve_mix_host') result(rc)
      import
      type(c_ptr), value :: fleet, weight, acc_flux, acc_uu
      type(sbd_mix_in), intent(in) :: min
      type(sbd_batch_out), intent(in) :: bout
      integer(c_int) :: rc
    end function
    function sbd_fleet_gas_terms(fleet, gas, npoint, wl, lay, nch, nk, wt, failed, dtaug_out, lay_token) &
         bind(C, name='sbd_fleet_gas_terms') result(rc)
      import
      type(c_ptr), value :: fleet, wl, lay, nk, wt, failed, dtaug_out
      integer(c_int64_t), intent(out) :: lay_token
      type(sbd_gas_model), intent(in) :: gas
      integer(c_int32_t), value :: npoint, nch
      integer(c_int) :: rc
    end function
    function sbd_fleet_point_terms(fleet, gas, scat, npoint, wl, nch, nk, wt, failed, dtaug_out, lay_out, lay_token) &
         bind(C, name='sbd_fleet_point_terms') result(rc)
      import
      type(c_ptr), value :: fleet, wl, nk, wt, failed, dtaug_out, lay_out
      integer(c_int64_t), intent(out) :: lay_token
      type(sbd_gas_model), intent(in) :: gas
      type(sbd_scat_model), intent(in) :: scat
      integer(c_int32_t), value :: npoint, nch
      integer(c_int) :: rc
    end function
    function sbd_scatter_blocks_host(scat, npoint, wl, nch, lay_out) bind(C, name='sbd_scatter_blocks_host') result(rc)
      import
      type(sbd_scat_model), intent(in) :: scat
      type(c_ptr), value :: wl, lay_out
      integer(c_int32_t), value :: npoint, nch
      integer(c_int) :: rc
    end function
    function sbd_engine_nlevel(eng) bind(C, name='sbd_engine_nlevel') result(n)
      import
      type(c_ptr), value :: eng
      integer(c_int32_t) :: n
    end function
    function sbd_engine_chunk(eng) bind(C, name='sbd_engine_chunk') result(n)
      import
      type(c_ptr), value :: eng
      integer(c_int32_t) :: n
    end function
    function sbd_abi_version() bind(C, name='sbd_abi_version') result(n)
      import
      integer(c_int32_t) :: n
    end function
    function sbd_strerror_c(code) bind(C, name='sbd_strerror') result(p)
      import
      integer(c_int), value :: code
      type(c_ptr) :: p
    end function
    function sbd_last_error_c() bind(C, name='sbd_last_error') result(p)
      import
      type(c_ptr) :: p
    end function
    function c_strlen(p) bind(C, name='strlen') result(n)
      import
      type(c_ptr), value :: p
      integer(c_size_t) :: n
    end function
  end interface

contains

  function cstr(p) result(s)
    type(c_ptr), intent(in) :: p
    character(len=:), allocatable :: s
    character(kind=c_char), pointer :: f(:)
    integer :: n, i
    if (.not. c_associated(p)) then
      s = ''
      return
    end if
    n = int(c_strlen(p))
    call c_f_pointer(p, f, [n])
    allocate(character(len=n) :: s)
    do i = 1, n
      s(i:i) = f(i)
    end do
  end function

  function sbd_strerror_f(code) result(s)
    integer(c_int), intent(in) :: code
    character(len=:), allocatable :: s
    s = cstr(sbd_strerror_c(code))
  end function

  function sbd_last_error_f() result(s)
    character(len=:), allocatable :: s
    s = cstr(sbd_last_error_c())
  end function

end module sbd_engine_mod
