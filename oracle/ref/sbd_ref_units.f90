! TEST INFRASTRUCTURE (oracle/_ref recipe) -- not part of the product path.
!
! Building-block driver: runs individual *reference* routines (objects compiled
! from /root/reference by build_ref.sh) on inputs read from a stream file, so
! tests/test_oracle_units.py can pin the C restatement routine by routine.
!
!   ref_units_cli IN.bin OUT.bin
!   op 1: QGAUSN (disort.f:5984)   in: m               out: gmu(m), gwt(m)
!   op 2: PLKAVG (disort.f:5410)   in: lo, hi, t       out: value
!   op 3: ASYMTX (disort.f:873)    in: m, aa(m,m)      out: ier, eval(m), evec(m,m)
!   op 4: LEPOLY (disort.f:5286)   in: nmu, maxmu, twonm1, mu(nmu)
!                                  out: ylm(0:maxmu,nmu) after each m=0..twonm1
!   op 5: SGBCO+SGBSL (disutil.f:426,920) in: n, ml, mu, lda, abd(lda,n), b(n)
!                                  out: rcond, x(n)
function bdref(wvnmlo, wvnmhi, mu, mup, dphi)
  implicit none
  integer, parameter :: dp = selected_real_kind(10)
  real(dp) :: bdref, wvnmlo, wvnmhi, mu, mup, dphi
  bdref = 0
  stop 'ref_units_cli: BDREF is out of scope'
end function

program sbd_ref_units
  implicit none
  integer, parameter :: dp = selected_real_kind(10)
  character(len=1024) :: fin, fout
  integer :: op, m, n, i, ier, nmu, mm, twonm1, maxmu, ml, mu_, lda
  real(dp), allocatable :: a(:,:), ev(:,:), eval(:), wk(:), gmu(:), gwt(:), &
       mu(:), ylm(:,:), sqt(:), b(:), z(:)
  integer, allocatable :: ipvt(:)
  real(dp) :: lo, hi, t, r, rcond
  real(dp), external :: plkavg
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(21, file=trim(fin), access='stream', form='unformatted', status='old')
  open(22, file=trim(fout), access='stream', form='unformatted', status='replace')
  open(11, status='scratch', form='formatted')
  do
    read(21, end=99) op
    select case (op)
    case (1)
      read(21) m
      allocate(gmu(m), gwt(m))
      call qgausn(m, gmu, gwt)
      write(22) gmu, gwt
      deallocate(gmu, gwt)
    case (2)
      read(21) lo, hi, t
      r = plkavg(lo, hi, t)
      write(22) r
    case (3)
      read(21) m
      allocate(a(m,m), ev(m,m), eval(m), wk(2*m))
      read(21) a
      call asymtx(a, ev, eval, m, m, m, ier, wk)
      write(22) real(ier, dp), eval, ev
      deallocate(a, ev, eval, wk)
    case (4)
      read(21) nmu, maxmu, twonm1
      allocate(mu(nmu), ylm(0:maxmu, nmu), sqt(1000))
      read(21) mu
      do i = 1, 1000
        sqt(i) = sqrt(float(i))      ! as disort.f:452-454
      end do
      ylm = 0
      do mm = 0, twonm1
        call lepoly(nmu, mm, maxmu, twonm1, mu, sqt, ylm)
        write(22) ylm
      end do
      deallocate(mu, ylm, sqt)
    case (5)
      read(21) n, ml, mu_, lda
      allocate(a(lda, n), b(n), z(n), ipvt(n))
      read(21) a, b
      rcond = 0
      call sgbco(a, lda, n, ml, mu_, ipvt, rcond, z)
      call sgbsl(a, lda, n, ml, mu_, ipvt, b, 0)
      write(22) rcond, b
      deallocate(a, b, z, ipvt)
    case default
      stop 'ref_units_cli: bad op'
    end select
  end do
99 continue
  close(22)
end program
