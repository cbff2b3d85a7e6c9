"""TEST INFRASTRUCTURE (like everything under oracle/): numpy restatement of what the reference does between its band
model and the DISORT call for one (wavelength, k-term) -- the statements the engine's assemble_kernel executes on the
device for a batch in compact form (sbd_mix_in, include/sbdart_amd.h, ABI v6):

  depthscl   taugas.f:7598-7603   dtau = ((gas + cloud) + aerosol) + Rayleigh;  wreal = scattering / dtau, 0 where dtau <= tiny
  GETMOM     disutil.f:2104-2209  iphas 1 isotropic, 2 Rayleigh (PMOM(2) = 0.1), 3 Henyey-Greenstein (PMOM(K) = GG**K)
  taucloud   taucloud.f:103, 132  a layer's cloud adds TAUCLD*WCLD*PMOM;  tauaero.f:1300, 1330  aerosols add PM*DTAUA*WAER
  normom     drt.f:1390-1395      + 0.1 * Rayleigh depth in the second moment, divided by the scattering depth, PMOM(0) = 1

GG**K with an INTEGER exponent is not pow(): the reference's compiler (like every Fortran compiler here) forms it by
square-and-multiply from the low bit of K (compiler-rt's __powidf2), and 0.1 is a REAL*4 literal -- both reproduced, so
that the arrays this module makes are bit-equal to the device's and to the band model's.
Pinned: tests/test_mix_assembly.py checks powi_fortran against the compiled reference's GETMOM, and `assemble` against
the Fortran host's work items (themselves bit-equal to the live reference: tests/test_band_model.py) for runs with
clouds, boundary-layer and stratospheric aerosols (SBD_DUMP_MIX) -- parity NOT otherwise unpinned."""
import numpy as np

RAY2 = float(np.float32(0.1))
TINY = np.finfo(np.float64).tiny


def powi_fortran(a, k):
    """a**k for integer k >= 0 as __powidf2 forms it (elementwise on arrays)."""
    a = np.array(a, dtype=np.float64, copy=True)
    r = np.ones_like(a)
    b = int(k)
    while True:
        if b & 1:
            r = r * a
        b //= 2
        if b == 0:
            break
        a = a * a
    return r


def assemble(point_of, dtaug, lay, family, nmom):
    """(dtauc [W][L], ssalb [W][L], pmom [P][L][nmom+1]) of a compact batch: lay [P][4 + 3 nterm][L], family [nterm]."""
    point_of = np.asarray(point_of)
    lay = np.asarray(lay, dtype=np.float64)
    dc, da, dr, scat = lay[:, 0], lay[:, 1], lay[:, 2], lay[:, 3]
    dtauc = ((np.asarray(dtaug, dtype=np.float64) + dc[point_of]) + da[point_of]) + dr[point_of]
    with np.errstate(divide="ignore", invalid="ignore"):
        ssalb = np.where(dtauc > TINY, scat[point_of] / dtauc, 0.0)
    P, L = scat.shape
    pmom = np.zeros((P, L, nmom + 1))
    pmom[:, :, 0] = 1.0
    for k in range(1, nmom + 1):
        q = np.zeros((P, L))
        for t, fam in enumerate(family):
            g, m1, m2 = lay[:, 4 + 3 * t], lay[:, 5 + 3 * t], lay[:, 6 + 3 * t]
            if fam == 3:
                pk = powi_fortran(g, k)
            elif fam == 2 and k == 2:
                pk = np.full((P, L), RAY2)
            else:
                pk = np.zeros((P, L))
            q = q + (pk * m1) * m2
        if k == 2:
            q = q + RAY2 * dr
        with np.errstate(divide="ignore", invalid="ignore"):
            pmom[:, :, k] = np.where(scat != 0.0, q / scat, q)
    return dtauc, ssalb, pmom
