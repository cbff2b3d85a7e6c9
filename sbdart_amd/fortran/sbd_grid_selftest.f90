! CPU-only check of the host's spectral-grid object against an optics file:
!   sbd_grid_selftest OPTICS.sbdrec wlinf wlsup wlinc
! prints nwl, the record count and the largest relative disagreement of wl/wvnmlo/wvnmhi with the records.
program sbd_grid_selftest
  use sbd_grid_mod
  use sbd_io_mod
  implicit none
  type(optics_t), allocatable :: recs(:)
  type(spectral_grid) :: g
  integer :: nrec, i
  character(len=1024) :: a
  real(kr) :: wlinf, wlsup, wlinc, wl, lo, hi, worst
  call get_command_argument(1, a)
  call read_optics(trim(a), recs, nrec)
  call get_command_argument(2, a); read(a, *) wlinf
  call get_command_argument(3, a); read(a, *) wlsup
  call get_command_argument(4, a); read(a, *) wlinc
  g = new_grid(wlinf, wlsup, wlinc)
  worst = 0
  do i = 1, nrec
    call g%band(recs(i)%iwl - 1, wl, lo, hi)
    worst = max(worst, abs(wl - recs(i)%wl)/wl, abs(lo - recs(i)%wvnmlo)/lo, abs(hi - recs(i)%wvnmhi)/hi)
  end do
  write(*, '(i8,i8,es12.4)') g%n, nrec, worst
end program
