! Host-side logic of the wavelength loop that stays in Fortran (SURVEY.md 8a rows a1-a3):
! spectral grid (setfilt's grid size, spectra.f:3370-3384; wllimits, drt.f:1657-1740),
! radiance viewing angles (vuangles, drt.f:813-889), the optical-property record reader
! ("SBDREC1", sbdart_amd/records.py), stdout0/1/2 (drt.f:892-1165) and the
! SBDART_WARNING.NN writer (errmsg, disutil.f:278-325).
module sbd_host_mod
  use iso_c_binding
  implicit none
  integer, parameter :: kr = selected_real_kind(10)
  integer, parameter :: mxly = 65, nstrms = 40        ! params.f:9-11
  real(kr), parameter :: zip = -1._kr

  type optics_t      ! one (wavelength, k-term) work item as handed to DISORT (drt.f:541-546)
    integer :: nlyr, nstr, nmom, numu, nphi, flags, kd, nk, iwl
    real(kr) :: wl, wt, ff, wvnmlo, wvnmhi, fbeam, umu0, phi0, albedo, btemp, ttemp, temis, fisot
    real(kr), allocatable :: dtauc(:), ssalb(:), temper(:), pmom(:,:), umu(:), phi(:)
  end type

  ! stdout1/stdout2 accumulators (outblk, drt.f:1-18)
  real(kr) :: topdn = 0, topup = 0, topdir = 0, botdn = 0, botup = 0, botdir = 0, phidw = 0
  real(kr) :: weq = 0, wfull = 0
  real(kr), allocatable :: uurs(:,:), fxdn(:), fxup(:), fxdir(:)

contains

  ! ---- spectral grid -------------------------------------------------------------
  integer function grid_size(wlmin, wlmax, wlinc) result(nwl)   ! spectra.f:3370-3384
    real(kr), intent(in) :: wlmin, wlmax
    real(kr), intent(inout) :: wlinc
    if (wlinc > 1._kr) then
      nwl = int(((10000._kr/wlmin) - (10000._kr/wlmax))/wlinc + 1._kr)
    else if (wlinc < 0._kr) then
      nwl = int(1 + log(wlmax/wlmin)/abs(wlinc))
    else
      if (wlinc == 0._kr) wlinc = (wlmax - wlmin)/max(10, 1 + int((wlmax - wlmin)/real(0.005, kr)))
      nwl = nint((wlmax - wlmin)/wlinc) + 1
    end if
    if (wlmin /= wlmax .and. nwl == 1) nwl = 2
  end function

  ! wavelength and band edges of spectral point il (0-based), drt.f:1657-1740
  subroutine wl_limits(il, nwl, wlinc, wl1, wl2, wl, wvnmlo, wvnmhi)
    integer, intent(in) :: il, nwl
    real(kr), intent(in) :: wlinc, wl1, wl2
    real(kr), intent(out) :: wl, wvnmlo, wvnmhi
    real(kr) :: wi, ww1, ww2, half
    wi = real(real(il), kr)          ! wi=float(il)
    half = 0.5_kr
    if (wlinc > 1._kr) then          ! equal increments of wavenumber
      wl = wlwn(wi); ww1 = wlwn(wi - half); ww2 = wlwn(wi + half)
    else if (wlinc < 0._kr) then     ! equal increments of log wavelength
      wl = wlln(wi); ww1 = wlln(wi - half); ww2 = wlln(wi + half)
    else                             ! equal increments of wavelength
      wl = wl1 + wi*wlinc
      ww1 = wl - half*wlinc
      ww2 = wl + half*wlinc
    end if
    if (il == 0 .and. il /= nwl - 1) ww1 = wl
    if (il == nwl - 1 .and. il /= 0) ww2 = wl
    if (ww1 == wl .and. ww2 == wl) then
      ww1 = wl - real(0.0005, kr)    ! single-precision literal in the reference
      ww2 = wl + real(0.0005, kr)
    end if
    wvnmlo = 10000._kr/ww2
    wvnmhi = 10000._kr/ww1
  contains
    real(kr) function wlwn(x)
      real(kr), intent(in) :: x
      real(kr) :: xx
      xx = x/(nwl - 1)
      wlwn = wl1*wl2/((1._kr - xx)*wl2 + xx*wl1)
    end function
    real(kr) function wlln(x)
      real(kr), intent(in) :: x
      real(kr) :: xx
      xx = x/(nwl - 1)
      wlln = wl1*(wl2/wl1)**xx
    end function
  end subroutine

  integer function numset(flag, a, n)     ! drt.f:1169-1179: index of the last entry /= flag
    real(kr), intent(in) :: flag, a(*)
    integer, intent(in) :: n
    integer :: i
    numset = 0
    do i = 1, n
      if (a(i) /= flag) numset = i
    end do
  end function

  ! radiance viewing angles, drt.f:813-889
  subroutine view_angles(nphi, phi, nzen, uzen, vzen, iout, nstr)
    integer, intent(inout) :: nphi, nzen, nstr
    integer, intent(in) :: iout
    real(kr), intent(inout) :: phi(nstrms), uzen(nstrms), vzen(nstrms)
    real(kr) :: p1, p2, z1, z2, xxx
    integer :: i, ii, nvzen
    if (nphi > 0) then
      p1 = min(phi(1), phi(2)); p2 = max(phi(1), phi(2))
      do i = 1, nphi
        phi(i) = p1 + (i - 1)*(p2 - p1)/real(real(nphi - 1), kr)
      end do
    else
      nphi = numset(zip, phi, nstrms)
      if (nphi == 0) then
        nphi = 19; p1 = 0; p2 = 180
        do i = 1, nphi
          phi(i) = p1 + (i - 1)*(p2 - p1)/real(real(nphi - 1), kr)
        end do
      end if
    end if
    nvzen = numset(90._kr, vzen, nstrms)
    do i = 1, nvzen
      uzen(i) = 180._kr - vzen(i)
    end do
    if (nzen > 0) then
      z1 = min(uzen(1), uzen(2)); z2 = max(uzen(1), uzen(2))
      ii = 0
      do i = 1, nzen
        xxx = z1 + (i - 1)*(z2 - z1)/real(real(nzen - 1), kr)
        if (abs(xxx - 90._kr) > real(0.05, kr)) then
          ii = ii + 1
          uzen(ii) = xxx
        end if
      end do
      nzen = ii
    else
      nzen = numset(zip, uzen, nstrms)
      if (nzen == 0) then
        select case (iout)
        case (5, 20); nzen = 18; z1 = 0;  z2 = 85
        case (6, 21); nzen = 18; z1 = 95; z2 = 180
        case default; nzen = 36; z1 = 0;  z2 = 180
        end select
        do i = 1, nzen
          uzen(i) = z1 + (z2 - z1)*(i - 1)/real(real(nzen - 1), kr)
        end do
        if (nstr == 4) nstr = min(2*(max(nphi, nzen)/2), nstrms)
      end if
    end if
  end subroutine

  ! ---- optical-property records --------------------------------------------------
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
      if (allocated(r%dtauc)) deallocate(r%dtauc, r%ssalb, r%temper, r%pmom, r%umu, r%phi)
      allocate(r%dtauc(r%nlyr), r%ssalb(r%nlyr), r%temper(0:r%nlyr), r%pmom(0:r%nmom, r%nlyr), &
               r%umu(r%numu), r%phi(r%nphi))
      read(u) r%dtauc, r%ssalb, r%temper, r%pmom, r%umu, r%phi
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

  ! ---- output (drt.f:892-1165) ---------------------------------------------------
  subroutine stdout0(iout, nwl, nz)
    integer, intent(in) :: iout, nwl, nz
    select case (iout)
    case (1, 5, 6)
      write(*, '(/,a)') '"tbf'
      write(*, '(i15)') nwl
    case (7)
      write(*, '(/,a)') '"fzw'
      write(*, '(i15)') nz
    end select
  end subroutine

  ! rfldir/rfldn/flup: fluxes at ALL levels 1..nz+1 or only (ntop,nbot) -> indices it, ib
  subroutine stdout1(nz, zlev, it, ib, iout, wl, dwl, wt, rfldir, rfldn, flup, ff, nphi, nzen, &
                     phi, uzen, uur, ju_top, ju_bot, kd, nk)
    integer, intent(in) :: nz, it, ib, iout, nphi, nzen, ju_top, ju_bot, kd, nk
    real(kr), intent(in) :: zlev(*), wl, dwl, wt, rfldir(*), rfldn(*), flup(*), ff, phi(*), uzen(*)
    real(kr), intent(in) :: uur(:,:,:)       ! (nzen, nlev_out, nphi)
    real(kr) :: dwt
    integer :: i, j, k, im
    dwt = wt*ff
    if (iout == 1 .or. iout == 5 .or. iout == 6) then
      if (kd == 1) then
        topdn = 0; topup = 0; topdir = 0; botdn = 0; botup = 0; botdir = 0; weq = 0; wfull = 0
      end if
      topdn  = topdn  + (rfldn(it) + rfldir(it))*dwt
      topup  = topup  + flup(it)*dwt
      topdir = topdir + rfldir(it)*dwt
      botdn  = botdn  + (rfldn(ib) + rfldir(ib))*dwt
      botup  = botup  + flup(ib)*dwt
      botdir = botdir + rfldir(ib)*dwt
      if (kd == nk) then
        weq = weq + dwl*ff
        wfull = wfull + dwl
      end if
      if (kd == nk) then
        if (weq == 0._kr) weq = real(1.e-30, kr)
        write(*, '(f12.8,f9.5,6es12.4)') wl, weq/wfull, real(topdn/weq), real(topup/weq), &
             real(topdir/weq), real(botdn/weq), real(botup/weq), real(botdir/weq)
      end if
      if (iout == 5 .or. iout == 6) then
        j = ju_top
        if (iout == 6) j = ju_bot
        if (kd == 1) uurs(1:nzen, 1:nphi) = 0
        do k = 1, nphi
          do i = 1, nzen
            uurs(i, k) = uurs(i, k) + uur(i, j, k)*dwt
          end do
        end do
        if (kd == nk) then
          write(*, '(3i4)') nphi, nzen
          write(*, '(10es12.4)') (real(phi(j)), j = 1, nphi)
          write(*, '(10es12.4)') (real(uzen(j)), j = 1, nzen)
          do i = nzen, 1, -1
            write(*, '(10es12.4)') (real(uurs(i, k)/weq), k = 1, nphi)
          end do
        end if
      end if
    end if
    if (any(iout == (/10, 11, 20, 21, 22, 23/)) .and. kd == nk) phidw = phidw + dwl*ff
    if (iout == 7 .or. iout == 11) then      ! needs all levels: rfldir(1:nz+1)
      if (iout == 7 .and. kd == 1) then
        fxdn(1:nz) = 0; fxup(1:nz) = 0; fxdir(1:nz) = 0
      end if
      do i = 1, nz
        fxdn(i) = fxdn(i) + (rfldn(i + 1) + rfldir(i + 1))*dwt
        fxup(i) = fxup(i) + flup(i + 1)*dwt
        fxdir(i) = fxdir(i) + rfldir(i + 1)*dwt
      end do
    end if
    if (iout == 7 .and. kd == nk) then
      write(*, '(//,f12.8)') wl
      write(*, '(/(10es11.3))') (zlev(im), im = nz, 1, -1)
      write(*, '(/(10es11.3))') (real(fxdir(i)), i = 1, nz)
      write(*, '(/(10es11.3))') (real(fxdn(i) - fxdir(i)), i = 1, nz)
      write(*, '(/(10es11.3))') (real(fxdn(i)), i = 1, nz)
      write(*, '(/(10es11.3))') (real(fxup(i)), i = 1, nz)
    end if
    if (any(iout == (/10, 20, 21, 23/))) then
      topdn  = topdn  + (rfldn(it) + rfldir(it))*dwt
      topup  = topup  + flup(it)*dwt
      topdir = topdir + rfldir(it)*dwt
      botdn  = botdn  + (rfldn(ib) + rfldir(ib))*dwt
      botup  = botup  + flup(ib)*dwt
      botdir = botdir + rfldir(ib)*dwt
    end if
    if (iout == 20 .or. iout == 21) then
      j = ju_top
      if (iout == 21) j = ju_bot
      do k = 1, nphi
        do i = 1, nzen
          uurs(i, k) = uurs(i, k) + uur(i, j, k)*dwt
        end do
      end do
    end if
    if (iout == 23) then
      do k = 1, nphi
        do i = 1, nzen
          if (uzen(nzen - i + 1) < 90._kr) then
            j = ju_top
          else
            j = ju_bot
          end if
          uurs(i, k) = uurs(i, k) + uur(i, j, k)*dwt
        end do
      end do
    end if
  end subroutine

  subroutine stdout2(iout, wlinf, wlsup, nphi, nzen, phi, uzen)
    integer, intent(in) :: iout, nphi, nzen
    real(kr), intent(in) :: wlinf, wlsup, phi(*), uzen(*)
    integer :: i, j, k
    if (iout == 10 .or. iout == 20 .or. iout == 21 .or. iout == 23) &
      write(*, '(3f11.4,6es12.4)') wlinf, wlsup, phidw, real(topdn), real(topup), real(topdir), &
           real(botdn), real(botup), real(botdir)
    if (iout == 20 .or. iout == 21 .or. iout == 23) then
      write(*, '(3i4)') nphi, nzen
      write(*, '(10es12.4)') (phi(j), j = 1, nphi)
      write(*, '(10es12.4)') (uzen(j), j = 1, nzen)
      do i = nzen, 1, -1
        write(*, '(20es12.4)') (real(uurs(i, k)), k = 1, nphi)
      end do
    end if
  end subroutine

  ! errmsg (disutil.f:278-325): message + copy of INPUT into SBDART_WARNING.NN, once per number
  subroutine warn_file(msgnum, messag)
    integer, intent(in) :: msgnum
    character(len=*), intent(in) :: messag
    logical, save :: issued(0:20) = .false.
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
    if (msgnum == 0) stop
    issued(msgnum) = .true.
  end subroutine

end module sbd_host_mod
