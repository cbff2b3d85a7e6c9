! Files around the engine: the per-work-item optical properties ("SBDREC1", the single definition of
! the format is sbdart_amd/records.py), the optional atmosphere levels (altitude and pressure, needed by
! the IOUT formats that print them and by ZOUT), and the reference's warning files.
module sbd_io_mod
  use sbd_grid_mod, only: kr
  implicit none
  private
  public :: optics_t, read_optics, write_optics, read_atmosphere, write_atmosphere, warn_file, warn_reset, fatal
  logical, save :: issued(0:20) = .false.              ! warning numbers written by this run (errmsg writes each once)

  ! the layer arrays of a work item read from an optics FILE (SBD_OPTICS); the band model's items keep theirs in its batch
  ! arrays, and their records carry ONE unallocated component instead of six array descriptors (75 001 records of the
  ! six-descriptor type cost the band model 7 of its 10.5 ms, round 6)
  type optics_arrays
    real(kr), allocatable :: dtauc(:), ssalb(:), temper(:), pmom(:,:), umu(:), phi(:)
  end type
  type optics_t      ! one (wavelength, k-term) work item as handed to DISORT (drt.f:541-546)
    integer :: nlyr, nstr, nmom, numu, nphi, flags, kd, nk, iwl
    real(kr) :: wl, wt, ff, wvnmlo, wvnmhi, fbeam, umu0, phi0, albedo, btemp, ttemp, temis, fisot
    type(optics_arrays), allocatable :: a
    ! bidirectional surface (LAMBER flag off): model 1 ocean, 2 Hapke, 3 Ross-Li, its run parameters, and the
    ! ocean's per-wavelength constants nr, ni, rsw (records.py: the block behind PHI when hdr(11) /= 0)
    integer :: ibdrf = 0
    real(kr) :: bpar(8) = 0, bitem(4) = 0
    ! KDIST = -1: sub-band ib (counting down from nb to 1) of a spectral point of the k-distribution file; the
    ! file's equivalent-width factor multiplies the filter value (drt.f:461, taugas.f:7695-7835)
    integer :: ib = 1, nb = 1
    real(kr) :: ewcoef = 1
  end type

contains

  subroutine fatal(msg)
    character(len=*), intent(in) :: msg
    write(0, '(a)') 'sbdart_amd: '//msg
    stop 1
  end subroutine

  subroutine read_optics(path, recs, nrec)
    character(len=*), intent(in) :: path
    type(optics_t), allocatable, intent(out) :: recs(:)
    integer, intent(out) :: nrec
    character(len=8) :: magic
    integer :: hdr(12), ohdr(4), has_out, nmax, ios, i, u
    real(kr) :: sc(16)
    real(kr), allocatable :: skip(:)
    type(optics_t) :: r
    type(optics_t), allocatable :: tmp(:)
    open(newunit=u, file=path, access='stream', form='unformatted', status='old', iostat=ios)
    if (ios /= 0) then
      write(0, '(a)') 'sbdart_amd: cannot open optics file '//trim(path)
      stop 2
    end if
    read(u) magic, nmax, has_out
    if (magic(1:7) /= 'SBDREC1') stop 'sbdart_amd: optics file is not SBDREC1'
    if (nmax < 0) nmax = huge(1)
    allocate(recs(256))
    nrec = 0
    do i = 1, nmax
      read(u, iostat=ios) hdr, sc
      if (ios /= 0) exit
      r%nlyr = hdr(1); r%nstr = hdr(2); r%nmom = hdr(3); r%numu = hdr(4); r%nphi = hdr(5)
      r%flags = hdr(6); r%kd = hdr(7); r%nk = hdr(8); r%iwl = hdr(9)
      r%wl = sc(1); r%wt = sc(2); r%ff = sc(3); r%wvnmlo = sc(4); r%wvnmhi = sc(5); r%fbeam = sc(6)
      r%umu0 = sc(7); r%phi0 = sc(8); r%albedo = sc(9); r%btemp = sc(10); r%ttemp = sc(11)
      r%temis = sc(12); r%fisot = sc(13)
      if (allocated(r%a)) deallocate(r%a)
      allocate(r%a)
      allocate(r%a%dtauc(r%nlyr), r%a%ssalb(r%nlyr), r%a%temper(0:r%nlyr), r%a%pmom(0:r%nmom, r%nlyr), &
               r%a%umu(r%numu), r%a%phi(r%nphi))
      read(u) r%a%dtauc, r%a%ssalb, r%a%temper, r%a%pmom, r%a%umu, r%a%phi
      r%ibdrf = hdr(11); r%bpar = 0; r%bitem = 0
      r%ib = 1; r%nb = 1
      if (hdr(12) /= 0) then
        r%ib = iand(hdr(12), 65535); r%nb = ishft(hdr(12), -16)
      end if
      if (r%ibdrf /= 0) read(u) r%bpar, r%bitem
      if (has_out /= 0) then       ! reference outputs, if present, are ignored by the host
        read(u) ohdr
        allocate(skip(5*ohdr(2)))
        read(u) skip
        deallocate(skip)
        if (iand(r%flags, 2) == 0) then
          allocate(skip(ohdr(3)*ohdr(2)*r%nphi))
          read(u) skip
          deallocate(skip)
        end if
      end if
      nrec = nrec + 1
      if (nrec > size(recs)) then
        allocate(tmp(2*size(recs)))
        tmp(1:nrec - 1) = recs(1:nrec - 1)
        call move_alloc(tmp, recs)
      end if
      recs(nrec) = r
    end do
    close(u)
  end subroutine

  ! the work items as an input-only SBDREC1 file (what read_optics reads)
  ! (the layer arrays either inside the records or, when the band model made them, in its batch arrays -- the moments
  !  there one block per wavelength, bpmom(:, :, iwl))
  subroutine write_optics(path, recs, nrec, bdtauc, bssalb, bpmom, btemper, umu, phi)
    character(len=*), intent(in) :: path
    type(optics_t), intent(in) :: recs(:)
    integer, intent(in) :: nrec
    real(kr), intent(in), optional :: bdtauc(:, :), bssalb(:, :), bpmom(:, :, :), btemper(:), umu(:), phi(:)
    integer :: u, i, hdr(12)
    real(kr) :: sc(16)
    open(newunit=u, file=path, access='stream', form='unformatted', status='replace')
    write(u) 'SBDREC1'//achar(0), nrec, 0
    do i = 1, nrec
      hdr = 0; sc = 0
      hdr(1:9) = (/recs(i)%nlyr, recs(i)%nstr, recs(i)%nmom, recs(i)%numu, recs(i)%nphi, recs(i)%flags, &
                   recs(i)%kd, recs(i)%nk, recs(i)%iwl/)
      hdr(11) = recs(i)%ibdrf
      if (recs(i)%nb > 1) hdr(12) = recs(i)%ib + ishft(recs(i)%nb, 16)
      sc(1:13) = (/recs(i)%wl, recs(i)%wt, recs(i)%ff, recs(i)%wvnmlo, recs(i)%wvnmhi, recs(i)%fbeam, recs(i)%umu0, &
                   recs(i)%phi0, recs(i)%albedo, recs(i)%btemp, recs(i)%ttemp, recs(i)%temis, recs(i)%fisot/)
      write(u) hdr, sc
      if (present(bdtauc)) then
        write(u) bdtauc(:, i), bssalb(:, i), btemper, bpmom(:recs(i)%nmom + lbound(bpmom, 1), :, recs(i)%iwl), umu, phi
      else
        write(u) recs(i)%a%dtauc, recs(i)%a%ssalb, recs(i)%a%temper, recs(i)%a%pmom, recs(i)%a%umu, recs(i)%a%phi
      end if
      if (recs(i)%ibdrf /= 0) write(u) recs(i)%bpar, recs(i)%bitem
    end do
    close(u)
  end subroutine

  ! Atmosphere levels, text: first line nz, then nz lines "z[km] p[mb]" from the SURFACE upwards
  ! (z(1) is the bottom, like the reference's profile arrays).  found = .false. when there is no file.
  subroutine read_atmosphere(path, nz, z, p, found)
    character(len=*), intent(in) :: path
    integer, intent(in) :: nz
    real(kr), intent(out) :: z(nz), p(nz)
    logical, intent(out) :: found
    integer :: u, ios, n, i
    found = .false.
    z = 0; p = 0
    open(newunit=u, file=path, status='old', form='formatted', iostat=ios)
    if (ios /= 0) return
    read(u, *, iostat=ios) n
    if (ios /= 0 .or. n /= nz) call fatal('atmosphere file '//trim(path)//' does not hold NZ levels of this run')
    do i = 1, nz
      read(u, *, iostat=ios) z(i), p(i)
      if (ios /= 0) call fatal('atmosphere file '//trim(path)//' is truncated')
    end do
    close(u)
    found = .true.
  end subroutine

  ! errmsg (disutil.f:278-325): message + copy of INPUT into SBDART_WARNING.NN, once per number;
  ! number 0 is fatal
  subroutine warn_reset()                                ! a new run of a batch starts with no warning issued
    issued = .false.
  end subroutine

  subroutine write_atmosphere(path, nz, z, p)
    character(len=*), intent(in) :: path
    integer, intent(in) :: nz
    real(kr), intent(in) :: z(nz), p(nz)
    integer :: u, i
    open(newunit=u, file=path, status='replace', form='formatted')
    write(u, '(i0)') nz
    do i = 1, nz
      write(u, '(2es26.17e3)') z(i), p(i)
    end do
    close(u)
  end subroutine

  ! (stops: number 0 ends the run -- the process too unless the caller runs a batch and says so)
  subroutine warn_file(msgnum, messag, stops)
    integer, intent(in) :: msgnum
    character(len=*), intent(in) :: messag
    logical, intent(in), optional :: stops
    character(len=2) :: num
    character(len=132) :: line
    integer :: u, v, ios
    if (msgnum > 0) then
      if (issued(msgnum)) return
      line = 'WARNING >>>>>'
    else
      line = 'ERROR  >>>>>>'
    end if
    write(num, '(i2.2)') msgnum
    open(newunit=u, file='SBDART_WARNING.'//num, status='unknown', form='formatted')
    write(u, '(a,1x,a)') trim(line), messag
    write(u, '(/70("#")/)')
    open(newunit=v, file='INPUT', status='old', iostat=ios)
    if (ios == 0) then
      do
        read(v, '(a)', iostat=ios) line
        if (ios /= 0) exit
        write(u, '(a)') trim(line)
      end do
      close(v)
    end if
    close(u)
    if (msgnum == 0) then
      if (present(stops)) then
        if (.not. stops) return
      end if
      stop
    end if
    issued(msgnum) = .true.
  end subroutine

end module sbd_io_mod
