"""TEST INFRASTRUCTURE (like everything under oracle/): numpy restatement of what the reference does between its band
model and the DISORT call for one (wavelength, k-term) -- the statements the engine's assemble_kernel executes on the
device for a batch in compact form (sbd_mix_in, include/sbdart_amd.h):

  depthscl   taugas.f:7625-7646   dtaus = gas + (cloud + aerosol + Rayleigh);  wreal = scattering / dtaus, 0 where dtaus is 0
  GETMOM     disutil.f:2176-2188  iphas 3 (Henyey-Greenstein): PMOM(K) = GG**K;  iphas 2 (Rayleigh): PMOM(2) = 0.1
  normom     drt.f:1366-1397      moments x scattering depth summed over the scatterers, divided by the total, PMOM(0) = 1

GG**K with an INTEGER exponent is not pow(): the reference's compiler (like every Fortran compiler here) forms it by
square-and-multiply from the low bit of K (compiler-rt's __powidf2), and 0.1 is a REAL*4 literal -- both reproduced, so
that the arrays this module makes are bit-equal to the device's and to the band model's for a Henyey-Greenstein scatterer.
Pinned: tests/test_mix_assembly.py checks powi_fortran against the compiled reference's GETMOM through the golden
records of sbchk.2 (cloud, imomc = 3) -- parity NOT otherwise unpinned."""
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


def assemble(point_of, dtaug, dtaux, tsc_hg, g_hg, tsc_ray, nmom):
    """(dtauc [W][L], ssalb [W][L], pmom [P][L][nmom+1]) of a compact batch."""
    point_of = np.asarray(point_of)
    dtauc = np.asarray(dtaug, dtype=np.float64) + np.asarray(dtaux, dtype=np.float64)[point_of]
    scat = np.asarray(tsc_hg, dtype=np.float64) + np.asarray(tsc_ray, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        ssalb = np.where(dtauc > TINY, scat[point_of] / dtauc, 0.0)
    P, L = scat.shape
    pmom = np.zeros((P, L, nmom + 1))
    pmom[:, :, 0] = 1.0
    for k in range(1, nmom + 1):
        q = tsc_hg * powi_fortran(g_hg, k)
        if k == 2:
            q = q + RAY2 * tsc_ray
        with np.errstate(divide="ignore", invalid="ignore"):
            pmom[:, :, k] = np.where(scat != 0.0, q / scat, q)
    return dtauc, ssalb, pmom
