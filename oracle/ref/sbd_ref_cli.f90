! TEST INFRASTRUCTURE (oracle/_ref recipe) -- not part of the product path.
!
! Command-line driver around the *reference* DISORT (disort.f:1-871), linked
! from objects compiled out of /root/reference by build_ref.sh:
!
!     disort_ref_cli IN.sbdrec OUT.sbdrec [NREPEAT]
!
! reads "SBDREC1" input records (layout: sbdart_amd/records.py), calls the
! reference DISORT once per record exactly as drt.f:541-546 does (usrtau=F,
! ibcnd=0, prnt=F), and writes the records back with the outputs appended.
! With NREPEAT>1 the whole list is re-solved NREPEAT times and the wall time
! of the DISORT calls alone is printed ("TIMING <solves> <seconds>") -- this is
! the `cpu_baseline.kind = "reference"` leg of bench.py.
!
! Only Lambertian surfaces are in scope, so BDREF (spectra.f:249) is stubbed:
! DISORT reaches it only when LAMBER is false.

module sbd_cli_types
  implicit none
  integer, parameter :: dp = selected_real_kind(10)
  type rec_t
    integer :: hdr(12)
    real(dp) :: sc(16)
    real(dp), allocatable :: dtauc(:), ssalb(:), temper(:), pmom(:,:), &
         umu(:), phi(:)
    integer :: ohdr(4)
    real(dp), allocatable :: flx(:,:), uu(:,:,:), albtrn(:,:)
  end type
end module

function bdref(wvnmlo, wvnmhi, mu, mup, dphi)
  use sbd_cli_types, only: dp
  implicit none
  real(dp) :: bdref, wvnmlo, wvnmhi, mu, mup, dphi
  bdref = 0
  stop 'sbd_ref_cli: BDREF called -- non-Lambertian surfaces are out of scope'
end function

program sbd_ref_cli
  use sbd_cli_types
  implicit none
  character(len=1024) :: fin, fout, arg
  character(len=8) :: magic
  character(len=127) :: header
  integer :: nrec, has_out, nrep, irep, i, n, ios, narg
  integer :: nlyr, nstr, nmom, numu, nphi, flags, mxumu, mxphi
  integer :: lc, k, lu, iu, j, ntau, numu_io, nstr_io, ibcnd
  integer(8) :: c0, c1, crate
  type(rec_t), allocatable :: recs(:)
  type(rec_t) :: tmp
  integer :: skip_ohdr(4)
  real(dp), allocatable :: skipbuf(:)
  logical :: plank, onlyfl, lamber, usrang, corint, prnt(5)
  real(dp), allocatable :: dtauc(:), ssalb(:), temper(:), pmom(:,:), umu(:), &
       phi(:), utau(:), rfldir(:), rfldn(:), flup(:), dfdt(:), uavg(:), &
       uu(:,:,:), albmed(:), trnmed(:)
  real(dp) :: secs
  external disort

  narg = command_argument_count()
  if (narg < 2) stop 'usage: disort_ref_cli IN.sbdrec OUT.sbdrec [NREPEAT]'
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  nrep = 1
  if (narg >= 3) then
    call get_command_argument(3, arg)
    read(arg, *) nrep
  end if

  open(21, file=trim(fin), access='stream', form='unformatted', status='old')
  read(21) magic, nrec, has_out
  if (magic(1:7) /= 'SBDREC1') stop 'bad magic'
  if (nrec < 0) nrec = 100000000
  allocate(recs(0))
  n = 0
  do i = 1, nrec
    read(21, iostat=ios) tmp%hdr, tmp%sc
    if (ios /= 0) exit
    nlyr = tmp%hdr(1); nmom = tmp%hdr(3); numu = tmp%hdr(4); nphi = tmp%hdr(5)
    if (allocated(tmp%dtauc)) deallocate(tmp%dtauc, tmp%ssalb, tmp%temper, &
         tmp%pmom, tmp%umu, tmp%phi)
    allocate(tmp%dtauc(nlyr), tmp%ssalb(nlyr), tmp%temper(0:nlyr), &
         tmp%pmom(0:nmom, nlyr), tmp%umu(numu), tmp%phi(nphi))
    read(21) tmp%dtauc, tmp%ssalb, tmp%temper, tmp%pmom, tmp%umu, tmp%phi
    if (has_out /= 0) then
      read(21) skip_ohdr
      allocate(skipbuf(5*skip_ohdr(2)))
      read(21) skipbuf
      deallocate(skipbuf)
      if (iand(tmp%hdr(6), 2) == 0) then
        allocate(skipbuf(skip_ohdr(3)*skip_ohdr(2)*nphi))
        read(21) skipbuf
        deallocate(skipbuf)
      end if
    end if
    n = n + 1
    if (n > size(recs)) call grow(recs, max(64, 2*size(recs)))
    recs(n) = tmp
  end do
  close(21)
  nrec = n

  ! errmsg (disutil.f:278-325) copies unit 11 (drt.f's INPUT) into its warning
  ! file and rewinds it: give it an empty scratch unit.
  open(11, status='scratch', form='formatted')
  prnt = .false.
  header = ' '
  secs = 0
  call system_clock(count_rate=crate)

  do irep = 1, nrep
    do i = 1, nrec
      nlyr = recs(i)%hdr(1); nstr = recs(i)%hdr(2); nmom = recs(i)%hdr(3)
      numu = recs(i)%hdr(4); nphi = recs(i)%hdr(5); flags = recs(i)%hdr(6)
      plank  = iand(flags, 1) /= 0
      onlyfl = iand(flags, 2) /= 0
      lamber = iand(flags, 4) /= 0
      usrang = iand(flags, 8) /= 0
      corint = iand(flags, 16) /= 0
      ibcnd = recs(i)%hdr(10)
      mxumu = max(2*numu, abs(nstr), 1)          ! (IBCND = 1 doubles the user angles in place, disort.f:2672-2687)
      mxphi = max(nphi, 1)
      allocate(dtauc(nlyr), ssalb(nlyr), temper(0:nlyr), pmom(0:nmom, nlyr), &
           umu(mxumu), phi(mxphi), utau(nlyr+1), rfldir(nlyr+1), &
           rfldn(nlyr+1), flup(nlyr+1), dfdt(nlyr+1), uavg(nlyr+1), &
           uu(mxumu, nlyr+1, mxphi), albmed(mxumu), trnmed(mxumu))
      dtauc = recs(i)%dtauc; ssalb = recs(i)%ssalb; temper = recs(i)%temper
      pmom = recs(i)%pmom
      umu = 0; phi = 0
      umu(1:numu) = recs(i)%umu
      phi(1:nphi) = recs(i)%phi
      ntau = 0
      numu_io = numu
      nstr_io = nstr
      call system_clock(c0)
      call disort(nlyr, dtauc, ssalb, corint, nmom, pmom, temper, &
           recs(i)%sc(4), recs(i)%sc(5), .false., ntau, utau, nstr_io, &
           usrang, numu_io, umu, nphi, phi, ibcnd, recs(i)%sc(6), recs(i)%sc(7), &
           recs(i)%sc(8), recs(i)%sc(13), lamber, recs(i)%sc(9), &
           recs(i)%sc(10), recs(i)%sc(11), recs(i)%sc(12), plank, onlyfl, &
           recs(i)%sc(14), prnt, header, nlyr, nlyr+1, mxumu, mxphi, nmom, &
           rfldir, rfldn, flup, dfdt, uavg, uu, albmed, trnmed)
      call system_clock(c1)
      secs = secs + real(c1 - c0, dp) / real(crate, dp)
      if (irep == 1) then
        recs(i)%ohdr = 0
        recs(i)%ohdr(1) = nstr_io
        recs(i)%ohdr(2) = nlyr + 1
        recs(i)%ohdr(3) = numu_io                  ! (USRANG off: DISORT set it to NSTR, disort.f:2655-2669)
        allocate(recs(i)%flx(nlyr+1, 5))
        recs(i)%flx(:,1) = rfldir; recs(i)%flx(:,2) = rfldn
        recs(i)%flx(:,3) = flup;   recs(i)%flx(:,4) = dfdt
        recs(i)%flx(:,5) = uavg
        if (.not. onlyfl) then
          allocate(recs(i)%uu(numu_io, nlyr+1, nphi))
          recs(i)%uu = uu(1:numu_io, 1:nlyr+1, 1:nphi)
        end if
        if (ibcnd == 1) then                       ! ALBMED, TRNMED at the (positive) output angles
          allocate(recs(i)%albtrn(numu_io, 2))
          recs(i)%albtrn(:, 1) = albmed(1:numu_io); recs(i)%albtrn(:, 2) = trnmed(1:numu_io)
        end if
      end if
      deallocate(dtauc, ssalb, temper, pmom, umu, phi, utau, rfldir, rfldn, &
           flup, dfdt, uavg, uu, albmed, trnmed)
    end do
  end do

  open(22, file=trim(fout), access='stream', form='unformatted', &
       status='replace')
  write(22) 'SBDREC1'//char(0), nrec, 1
  do i = 1, nrec
    write(22) recs(i)%hdr, recs(i)%sc
    write(22) recs(i)%dtauc, recs(i)%ssalb, recs(i)%temper, recs(i)%pmom, &
         recs(i)%umu, recs(i)%phi
    write(22) recs(i)%ohdr
    write(22) recs(i)%flx
    if (iand(recs(i)%hdr(6), 2) == 0) write(22) recs(i)%uu
    if (recs(i)%hdr(10) == 1) write(22) recs(i)%albtrn
  end do
  close(22)
  write(*, '(a,i10,es16.8)') 'TIMING ', nrec*nrep, secs

contains
  subroutine grow(a, newsize)
    type(rec_t), allocatable, intent(inout) :: a(:)
    integer, intent(in) :: newsize
    type(rec_t), allocatable :: b(:)
    integer :: m
    m = size(a)
    allocate(b(newsize))
    if (m > 0) b(1:m) = a(1:m)
    call move_alloc(b, a)
  end subroutine
end program sbd_ref_cli
