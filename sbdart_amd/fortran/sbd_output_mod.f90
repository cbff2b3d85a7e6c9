! The IOUT output formats of SBDART (drt.f:200-231 lists them; their text layout is what RunRT's
! readers and TestRuns/sbchk.* consume), table driven.
!
! Every format is described by one row of `formats`: whether it prints a record per spectral point
! or one per run, which accumulators it needs (boundary fluxes, the flux profile, radiances at the
! top / the bottom / split by hemisphere / at every level) and its banner.  One accumulator type
! (`spectral_sums`) holds stdout1's weighted sums (outblk, drt.f:1-18); it is filled either by a
! host loop over the k-terms of one spectral point (per-point formats: at most three items) or from
! the engine's reduced sums over the whole run (per-run formats: sbd_fleet_solve_host's acc_flux /
! acc_uu), and printed by the writers below with the reference's format strings.
module sbd_output_mod
  use sbd_grid_mod, only: kr
  implicit none
  private
  public :: iout_format, find_format, spectral_sums, sums_init, sums_clear, sums_add_item, &
            sums_from_engine, write_banner, write_point_record, write_run_record
  public :: rad_none, rad_top, rad_bottom, rad_split, rad_levels

  integer, parameter :: rad_none = 0, rad_top = 1, rad_bottom = 2, rad_split = 3, rad_levels = 4

  type :: iout_format
    integer :: code = 0
    logical :: per_point = .false.     ! a record per spectral point (else one per run)
    logical :: profile = .false.       ! fluxes at every level
    integer :: radiance = rad_none
    character(len=4) :: banner = ''    ! first line of the output, '' = none
  end type

  type(iout_format), parameter :: formats(10) = (/ &
       iout_format(1, .true., .false., rad_none, '"tbf'), iout_format(5, .true., .false., rad_top, '"tbf'), &
       iout_format(6, .true., .false., rad_bottom, '"tbf'), iout_format(7, .true., .true., rad_none, '"fzw'), &
       iout_format(10, .false., .false., rad_none, ''), iout_format(11, .false., .true., rad_none, ''), &
       iout_format(20, .false., .false., rad_top, ''), iout_format(21, .false., .false., rad_bottom, ''), &
       iout_format(22, .false., .true., rad_levels, ''), iout_format(23, .false., .false., rad_split, '') /)

  type :: spectral_sums
    integer :: nz = 0, nzen = 0, nphi = 0
    real(kr) :: down(2) = 0, up(2) = 0, direct(2) = 0      ! (1) top output level, (2) bottom output level
    real(kr) :: width_eq = 0, width_full = 0               ! sum of dwl*ff and of dwl over finished bands
    real(kr), allocatable :: prof_down(:), prof_up(:), prof_direct(:)   ! levels 2..nz+1, top-down
    real(kr), allocatable :: rad(:,:)                      ! (zenith, azimuth) at the format's level(s)
    real(kr), allocatable :: rad_lev(:,:,:)                ! (azimuth, zenith, level 2..nz+1)
  end type

contains

  function find_format(iout, found) result(f)
    integer, intent(in) :: iout
    logical, intent(out) :: found
    type(iout_format) :: f
    integer :: i
    found = .false.
    do i = 1, size(formats)
      if (formats(i)%code == iout) then
        f = formats(i)
        found = .true.
      end if
    end do
  end function

  subroutine sums_init(s, f, nz, nzen, nphi)
    type(spectral_sums), intent(out) :: s
    type(iout_format), intent(in) :: f
    integer, intent(in) :: nz, nzen, nphi
    s%nz = nz; s%nzen = nzen; s%nphi = nphi
    if (f%profile) allocate(s%prof_down(nz), s%prof_up(nz), s%prof_direct(nz))
    if (f%radiance /= rad_none) allocate(s%rad(max(nzen, 1), max(nphi, 1)))
    if (f%radiance == rad_levels) allocate(s%rad_lev(max(nphi, 1), max(nzen, 1), nz))
    call sums_clear(s)
  end subroutine

  subroutine sums_clear(s)
    type(spectral_sums), intent(inout) :: s
    s%down = 0; s%up = 0; s%direct = 0
    s%width_eq = 0; s%width_full = 0
    if (allocated(s%prof_down)) then
      s%prof_down = 0; s%prof_up = 0; s%prof_direct = 0
    end if
    if (allocated(s%rad)) s%rad = 0
    if (allocated(s%rad_lev)) s%rad_lev = 0
  end subroutine

  ! which of the two output levels a viewing zenith belongs to for rad_split: looking down
  ! (uzen < 90) is seen from the top level, looking up from the bottom level
  pure integer function split_level(uzen_deg) result(which)
    real(kr), intent(in) :: uzen_deg
    which = merge(1, 2, uzen_deg < 90._kr)
  end function

  ! one work item, weight w = wt*ff.  flux(lev, c): c = 1 rfldir, 2 rfldn, 3 flup; lev runs over the
  ! engine's output levels; lt/lb = positions of the top/bottom output level in that list;
  ! uu(zenith-reversed index iu, lev, azimuth) as DISORT returns it (ascending umu).
  subroutine sums_add_item(s, f, w, flux, lt, lb, uu, uzen)
    type(spectral_sums), intent(inout) :: s
    type(iout_format), intent(in) :: f
    real(kr), intent(in) :: w, flux(:,:), uu(:,:,:), uzen(:)
    integer, intent(in) :: lt, lb
    integer :: lev(2), e, i, k, j
    lev = (/lt, lb/)
    do e = 1, 2
      s%down(e) = s%down(e) + (flux(lev(e), 2) + flux(lev(e), 1))*w
      s%up(e) = s%up(e) + flux(lev(e), 3)*w
      s%direct(e) = s%direct(e) + flux(lev(e), 1)*w
    end do
    if (f%profile) then
      do i = 1, s%nz
        s%prof_down(i) = s%prof_down(i) + (flux(i + 1, 2) + flux(i + 1, 1))*w
        s%prof_up(i) = s%prof_up(i) + flux(i + 1, 3)*w
        s%prof_direct(i) = s%prof_direct(i) + flux(i + 1, 1)*w
      end do
    end if
    select case (f%radiance)
    case (rad_top, rad_bottom, rad_split)
      do k = 1, s%nphi
        do i = 1, s%nzen
          j = lev(merge(1, 2, f%radiance == rad_top))
          if (f%radiance == rad_split) j = lev(split_level(uzen(s%nzen - i + 1)))
          s%rad(i, k) = s%rad(i, k) + uu(i, j, k)*w
        end do
      end do
    case (rad_levels)
      do j = 1, s%nz
        do k = 1, s%nphi
          do i = 1, s%nzen
            s%rad_lev(k, i, j) = s%rad_lev(k, i, j) + uu(i, j + 1, k)*w
          end do
        end do
      end do
    end select
  end subroutine

  ! the same sums from the engine's reduction over the whole run: acc_flux(lev, c) with c = 1..5 DISORT's
  ! flux outputs, acc_uu(iu, lev, azimuth)
  subroutine sums_from_engine(s, f, acc_flux, acc_uu, lt, lb, uzen)
    type(spectral_sums), intent(inout) :: s
    type(iout_format), intent(in) :: f
    real(kr), intent(in) :: acc_flux(:,:), acc_uu(:,:,:), uzen(:)
    integer, intent(in) :: lt, lb
    call sums_add_item(s, f, 1._kr, acc_flux(:, 1:3), lt, lb, acc_uu, uzen)
  end subroutine

  subroutine write_banner(f, nwl, nz)
    type(iout_format), intent(in) :: f
    integer, intent(in) :: nwl, nz
    if (len_trim(f%banner) == 0) return
    write(*, '(/,a)') trim(f%banner)
    if (f%profile) then
      write(*, '(i15)') nz
    else
      write(*, '(i15)') nwl
    end if
  end subroutine

  ! record of one spectral point (IOUT 1/5/6: spectral fluxes per um [+ radiances]; IOUT 7: flux profile)
  subroutine write_point_record(s, f, wl, z, phi, uzen)
    type(spectral_sums), intent(in) :: s
    type(iout_format), intent(in) :: f
    real(kr), intent(in) :: wl, z(:), phi(:), uzen(:)
    real(kr) :: weq
    integer :: i, k
    if (f%profile) then
      write(*, '(//,f12.8)') wl
      write(*, '(/(10es11.3))') (z(i), i = s%nz, 1, -1)
      write(*, '(/(10es11.3))') (real(s%prof_direct(i)), i = 1, s%nz)
      write(*, '(/(10es11.3))') (real(s%prof_down(i) - s%prof_direct(i)), i = 1, s%nz)
      write(*, '(/(10es11.3))') (real(s%prof_down(i)), i = 1, s%nz)
      write(*, '(/(10es11.3))') (real(s%prof_up(i)), i = 1, s%nz)
      return
    end if
    weq = s%width_eq
    if (weq == 0._kr) weq = real(1.e-30, kr)
    write(*, '(f12.8,f9.5,6es12.4)') wl, weq/s%width_full, &
         real(s%down(1)/weq), real(s%up(1)/weq), real(s%direct(1)/weq), &
         real(s%down(2)/weq), real(s%up(2)/weq), real(s%direct(2)/weq)
    if (f%radiance /= rad_none) then
      write(*, '(3i4)') s%nphi, s%nzen
      write(*, '(10es12.4)') (real(phi(k)), k = 1, s%nphi)
      write(*, '(10es12.4)') (real(uzen(i)), i = 1, s%nzen)
      do i = s%nzen, 1, -1
        write(*, '(10es12.4)') (real(s%rad(i, k)/weq), k = 1, s%nphi)
      end do
    end if
  end subroutine

  ! record of a whole run (IOUT 10/11/20/21/22/23)
  subroutine write_run_record(s, f, wlinf, wlsup, z, p, phi, uzen)
    type(spectral_sums), intent(in) :: s
    type(iout_format), intent(in) :: f
    real(kr), intent(in) :: wlinf, wlsup, z(:), p(:), phi(:), uzen(:)
    real(kr), parameter :: grav = real(9.80665, kr), cp = real(1004., kr)   ! params.f:21, drt.f:1104
    real(kr) :: net, net_above, z_above, p_above, zz, pp, dfdz, heat
    integer :: i, j, k
    select case (f%code)
    case (11)
      write(*, '(i4,es15.7)') s%nz, s%width_eq
      net_above = 0; z_above = 0; p_above = 0
      do i = 1, s%nz                    ! top-down; prof_*(nz) is the surface
        zz = z(s%nz - i + 1)
        pp = p(s%nz - i + 1)
        net = s%prof_down(i) - s%prof_up(i)
        dfdz = 0
        heat = 0
        if (i > 1) then                 ! flux divergence and heating rate (K/day) of the layer above
          dfdz = (net_above - net)/(z_above - zz)
          heat = .01*3600*24*grav*(net_above - net)/(cp*(pp - p_above))
        end if
        net_above = net; z_above = zz; p_above = pp
        write(*, '(10es12.4)') zz, pp, real(s%prof_down(i)), real(s%prof_up(i)), real(s%prof_direct(i)), &
             real(dfdz), real(heat)
      end do
    case (22)
      write(*, '(3i4,es12.4)') s%nphi, s%nzen, s%nz, s%width_eq
      write(*, '(10es12.4)') (phi(i), i = 1, s%nphi)
      write(*, '(10es12.4)') (uzen(j), j = 1, s%nzen)
      write(*, '(10es12.4)') (z(k), k = s%nz, 1, -1)
      write(*, '(10es12.4)') (real(s%prof_down(k)), k = 1, s%nz)
      write(*, '(10es12.4)') (real(s%prof_up(k)), k = 1, s%nz)
      write(*, '(10es12.4)') (real(s%prof_direct(k)), k = 1, s%nz)
      write(*, '(10es12.4)') (((real(s%rad_lev(i, j, k)), i = 1, s%nphi), j = s%nzen, 1, -1), k = 1, s%nz)
    case default
      write(*, '(3f11.4,6es12.4)') wlinf, wlsup, s%width_eq, &
           real(s%down(1)), real(s%up(1)), real(s%direct(1)), real(s%down(2)), real(s%up(2)), real(s%direct(2))
      if (f%radiance /= rad_none) then
        write(*, '(3i4)') s%nphi, s%nzen
        write(*, '(10es12.4)') (phi(j), j = 1, s%nphi)
        write(*, '(10es12.4)') (uzen(j), j = 1, s%nzen)
        do i = s%nzen, 1, -1
          write(*, '(20es12.4)') (real(s%rad(i, k)), k = 1, s%nphi)
        end do
      end if
    end select
  end subroutine

end module sbd_output_mod
