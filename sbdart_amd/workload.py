"""Seeded synthetic spectral sweeps for throughput runs and full-size property tests
("16-stream SW sweep").

One work item = one (wavelength, k-term) DISORT solve.  A sweep has `nwl` spectral
points; point i carries nk_i in {1, 3} k-distribution terms (SW average 2.67,
taugas.f:7441-7460 / BASELINE.md), so W = sum nk_i work items.  Layer optical
properties come from splitmix64(seed): DTAUC = exp(U(-9,1)) rescaled so that the
column optical depth stays <= 50, SSALB = U(0, 0.999999), PMOM_k = (1-r) g^k + r*Ray_k
with g = U(0, 0.9) (Henyey-Greenstein, GETMOM iphas=3, disutil.f:2186-2188) and the
Rayleigh moments (PMOM_2 = 0.1, disutil.f:2176-2180) mixed in with r = U(0,1).
Level temperatures are the US-62 profile SBDART uses for idatm=6.
"""
from __future__ import annotations

import dataclasses

import numpy as np

# US standard atmosphere 1962 level temperatures (K), top-down, 34 levels (33 layers),
# as handed to DISORT by the reference for idatm=6 (captured DISORT argument TEMPER, a
# table of single-precision literals widened to fp64).
US62_TEMPER = np.array([np.float32(x) for x in (
    210.0, 210.0, 219.7, 270.6, 264.2, 253.4, 236.5, 226.5, 221.6, 220.6, 219.6, 218.6, 217.6, 216.6, 216.6, 216.6, 216.6, 216.6, 216.6, 216.6, 216.6, 216.6, 216.8, 223.2, 229.7, 236.2, 242.7, 249.2, 255.7, 262.2, 268.7, 275.1, 281.6, 288.1)], dtype=np.float64)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n uniform doubles in [0,1) from the splitmix64 stream started at `seed`."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@dataclasses.dataclass
class Sweep:
    nlyr: int
    nstr: int
    nmom: int
    nwl: int
    nk: np.ndarray        # [nwl] int
    wl_of: np.ndarray     # [W] spectral-point index of each work item
    weight: np.ndarray    # [W] k-term weight * filter (stdout1's dwt)
    wl: np.ndarray        # [nwl] micrometres
    dtauc: np.ndarray     # [W, nlyr]
    ssalb: np.ndarray     # [W, nlyr]
    pmom: np.ndarray      # [W, nlyr, nmom+1]
    wvnmlo: np.ndarray    # [W]
    wvnmhi: np.ndarray
    fbeam: np.ndarray
    albedo: np.ndarray
    plank: np.ndarray     # [W] uint8
    temper: np.ndarray    # [nlyr+1]
    umu0: float
    btemp: float
    ttemp: float
    temis: float

    @property
    def nwork(self) -> int:
        return int(self.dtauc.shape[0])

    def points(self, lo: int, hi: int) -> "Sweep":
        """The spectral points lo..hi-1 of this sweep with all their k-terms: one shard of a sweep that is cut BETWEEN
        spectral points (sbd_shard_range over the points; strong scaling, north_star's split)."""
        a = int(np.searchsorted(self.wl_of, lo, side="left"))
        b = int(np.searchsorted(self.wl_of, hi, side="left"))
        cut = lambda x: np.ascontiguousarray(x[a:b])
        return dataclasses.replace(self, nwl=hi - lo, nk=self.nk[lo:hi], wl_of=cut(self.wl_of) - lo, weight=cut(self.weight),
                                   wl=self.wl[lo:hi], dtauc=cut(self.dtauc), ssalb=cut(self.ssalb), pmom=cut(self.pmom),
                                   wvnmlo=cut(self.wvnmlo), wvnmhi=cut(self.wvnmhi), fbeam=cut(self.fbeam),
                                   albedo=cut(self.albedo), plank=cut(self.plank))


def sw_sweep(nwl: int, nstr: int = 16, nlyr: int = 33, seed: int = 12345, wlinf: float = 0.25,
             wlsup: float = 4.0, albedo: float = 0.2, sza_deg: float = 30.0,
             thermal_above_um: float = 2.0, shard: int = 0) -> Sweep:
    """Synthetic short-wave sweep of `nwl` spectral points on a uniform wavelength grid.
    `shard` offsets the PRNG stream so that ranks of a multi-GPU run get disjoint work."""
    nmom = min(nstr + 2, 40)                       # drt.f:490-494
    seed = int(seed) + 1000003 * int(shard)
    u = splitmix64(seed, nwl)
    nk = np.where(u < 0.835, 3, 1).astype(np.int64)     # mean 2.67
    W = int(nk.sum())
    wl_of = np.repeat(np.arange(nwl), nk)
    kidx = np.concatenate([np.arange(k) for k in nk])
    # k-term weights: (1/3,1/3,1/3)-like normalised triples with the LOWTRAN flavour 0.6/0.3/0.1
    wtab = np.array([0.6, 0.3, 0.1])
    weight = np.where(nk[wl_of] == 1, 1.0, wtab[kidx])
    wl = wlinf + (wlsup - wlinf) * (np.arange(nwl) + 0.5) / nwl
    dwl = (wlsup - wlinf) / nwl
    wvnmlo = 1.0e4 / (wl + 0.5 * dwl)
    wvnmhi = 1.0e4 / (wl - 0.5 * dwl)
    r = splitmix64(seed ^ 0x5DEECE66D, W * (2 * nlyr + 2)).reshape(W, 2 * nlyr + 2)
    dtauc = np.exp(-9.0 + 10.0 * r[:, :nlyr])
    # stronger absorption for the higher k-terms, like a k-distribution
    dtauc *= (1.0 + 4.0 * kidx)[:, None]
    tot = dtauc.sum(axis=1, keepdims=True)
    dtauc *= np.minimum(1.0, 50.0 / tot)
    ssalb = 0.999999 * r[:, nlyr:2 * nlyr]
    g = 0.9 * r[:, 2 * nlyr]
    ray = r[:, 2 * nlyr + 1]
    k = np.arange(nmom + 1, dtype=np.float64)
    hg = g[:, None] ** k[None, :]
    rayk = np.zeros(nmom + 1)
    rayk[0], rayk[2] = 1.0, 0.1
    pm = (1.0 - ray)[:, None] * hg + ray[:, None] * rayk[None, :]
    # layer-to-layer variation of the phase function: blend towards Rayleigh with height
    lw = np.linspace(1.0, 0.2, nlyr)
    pmom = lw[None, :, None] * rayk[None, None, :] + (1.0 - lw)[None, :, None] * pm[:, None, :]
    pmom[:, :, 0] = 1.0
    plank = (wl > thermal_above_um)[wl_of].astype(np.uint8)
    if nlyr + 1 == len(US62_TEMPER):
        temper = US62_TEMPER.copy()
    else:
        temper = np.interp(np.linspace(0, 1, nlyr + 1), np.linspace(0, 1, len(US62_TEMPER)), US62_TEMPER)
    return Sweep(nlyr=nlyr, nstr=nstr, nmom=nmom, nwl=nwl, nk=nk, wl_of=wl_of, weight=weight, wl=wl,
                 dtauc=np.ascontiguousarray(dtauc), ssalb=np.ascontiguousarray(ssalb),
                 pmom=np.ascontiguousarray(pmom), wvnmlo=wvnmlo[wl_of], wvnmhi=wvnmhi[wl_of],
                 fbeam=np.ones(W), albedo=np.full(W, albedo), plank=plank, temper=temper,
                 umu0=float(np.cos(np.deg2rad(sza_deg))), btemp=float(temper[-1]),
                 ttemp=float(temper[0]), temis=0.0)


def sweep_to_records(sw: Sweep, idx):
    """SolveRecords for a subset of a sweep (for the oracle / reference CLI)."""
    from .records import F_LAMBER, F_ONLYFL, F_PLANK, SolveRecord
    out = []
    for i in idx:
        out.append(SolveRecord(
            nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom,
            flags=F_ONLYFL | F_LAMBER | (F_PLANK if sw.plank[i] else 0),
            wvnmlo=float(sw.wvnmlo[i]), wvnmhi=float(sw.wvnmhi[i]), fbeam=float(sw.fbeam[i]),
            umu0=sw.umu0, phi0=0.0, albedo=float(sw.albedo[i]), btemp=sw.btemp, ttemp=sw.ttemp,
            temis=sw.temis, dtauc=sw.dtauc[i], ssalb=sw.ssalb[i], temper=sw.temper, pmom=sw.pmom[i],
            wl=float(sw.wl[sw.wl_of[i]]), wt=float(sw.weight[i]), iwl=int(sw.wl_of[i]) + 1))
    return out


@dataclasses.dataclass
class MixSweep:
    """The sweep of SURVEY 8(d) in the COMPACT form of include/sbdart_amd.h (sbd_mix_in, ABI v6): per spectral point the
    layer block lay[point] = [dtauc, dtaua, dtaur, tsc, then (g, m1, m2) per scattering term][nlyr] -- here a cloud-like
    Henyey-Greenstein scatterer (g = U(0, 0.9)), an aerosol-like one (g = U(0.5, 0.8)) and Rayleigh scattering --, per work
    item the gas of its k-term."""
    nlyr: int
    nstr: int
    nmom: int
    nwl: int
    point_of: np.ndarray   # [W]
    weight: np.ndarray     # [W]
    dtaug: np.ndarray      # [W, nlyr]
    lay: np.ndarray        # [nwl, 4 + 3 nterm, nlyr]
    family: tuple          # GETMOM's iphas per term
    wvnmlo: np.ndarray     # [nwl]
    wvnmhi: np.ndarray
    fbeam: np.ndarray
    albedo: np.ndarray
    plank: np.ndarray      # [nwl] uint8
    temper: np.ndarray
    umu0: float
    btemp: float
    ttemp: float
    temis: float

    @property
    def nwork(self) -> int:
        return int(self.dtaug.shape[0])

    def points(self, lo: int, hi: int) -> "MixSweep":
        """Spectral points lo..hi-1 with their k-terms (a shard cut between points, like Sweep.points)."""
        a = int(np.searchsorted(self.point_of, lo, side="left"))
        b = int(np.searchsorted(self.point_of, hi, side="left"))
        pt = lambda x: np.ascontiguousarray(x[lo:hi])
        return dataclasses.replace(self, nwl=hi - lo, point_of=np.ascontiguousarray(self.point_of[a:b] - lo).astype(np.int32),
                                   weight=np.ascontiguousarray(self.weight[a:b]), dtaug=np.ascontiguousarray(self.dtaug[a:b]),
                                   lay=pt(self.lay), wvnmlo=pt(self.wvnmlo), wvnmhi=pt(self.wvnmhi), fbeam=pt(self.fbeam),
                                   albedo=pt(self.albedo), plank=pt(self.plank))

    def mix_args(self):
        """Positional arguments of DisortFleet.solve_mix."""
        return (self.point_of, self.dtaug, self.lay, self.family, self.wvnmlo, self.wvnmhi, self.fbeam, self.albedo, self.plank)

    def arrays(self):
        """(dtauc [W][L], ssalb [W][L], pmom [nwl][L][nmom+1]): DISORT's arguments of this sweep, the way the engine forms
        them on the device (include/sbdart_amd.h, sbd_mix_in) -- for runs that want them resident in HBM beforehand.
        (A numpy statement of assemble_kernel like oracle/mix_restatement.py, which the tests compare it with: the
        product does not import the oracle.)"""
        lay = self.lay
        dc, da, dr, scat = lay[:, 0], lay[:, 1], lay[:, 2], lay[:, 3]
        po = self.point_of
        dtauc = ((self.dtaug + dc[po]) + da[po]) + dr[po]
        with np.errstate(divide="ignore", invalid="ignore"):
            ssalb = np.where(dtauc > np.finfo(np.float64).tiny, scat[po] / dtauc, 0.0)
        pmom = np.zeros((self.nwl, self.nlyr, self.nmom + 1))
        pmom[:, :, 0] = 1.0
        ray2 = float(np.float32(0.1))
        for k in range(1, self.nmom + 1):
            q = np.zeros_like(scat)
            for t, fam in enumerate(self.family):
                g, m1, m2 = lay[:, 4 + 3 * t], lay[:, 5 + 3 * t], lay[:, 6 + 3 * t]
                if fam == 3:                     # g**k by square-and-multiply from the low bit (a Fortran integer power)
                    a, r, b = g.copy(), np.ones_like(g), k
                    while True:
                        if b & 1:
                            r = r * a
                        b //= 2
                        if b == 0:
                            break
                        a = a * a
                    pk = r
                else:
                    pk = np.full_like(g, ray2 if (fam == 2 and k == 2) else 0.0)
                q = q + (pk * m1) * m2
            if k == 2:
                q = q + ray2 * dr
            with np.errstate(divide="ignore", invalid="ignore"):
                pmom[:, :, k] = np.where(scat != 0.0, q / scat, q)
        return dtauc, ssalb, pmom

    def h2d_bytes(self) -> int:
        return int(sum(a.nbytes for a in (self.point_of, self.dtaug, self.lay, self.wvnmlo, self.wvnmhi, self.fbeam,
                                          self.albedo, self.plank, self.weight)))


def sw_sweep_mix(nwl: int, nstr: int = 16, nlyr: int = 33, seed: int = 12345, wlinf: float = 0.25, wlsup: float = 4.0,
                 albedo: float = 0.2, sza_deg: float = 30.0, thermal_above_um: float = 2.0, shard: int = 0) -> MixSweep:
    """Same sizes, k-term structure, wavelength grid and column optical depths as sw_sweep; the layer optical properties
    are drawn per SPECTRAL POINT (what a band model delivers): extinction of the scatterers exp(U(-9,1)) split into a
    cloud-like Henyey-Greenstein part (single-scattering albedo U(0, 0.999999), g = U(0, 0.9)), an aerosol-like one in the
    lowest third of the layers (a fifth of the particles' extinction there, single-scattering albedo 0.9, g = U(0.5, 0.8))
    and a Rayleigh part that grows with height; per work item the gas absorption of its k-term (stronger for the higher
    terms)."""
    base = sw_sweep(nwl=nwl, nstr=nstr, nlyr=nlyr, seed=seed, wlinf=wlinf, wlsup=wlsup, albedo=albedo, sza_deg=sza_deg,
                    thermal_above_um=thermal_above_um, shard=shard)
    W = base.nwork
    seed = int(seed) + 1000003 * int(shard)
    r = splitmix64(seed ^ 0x2545F4914F6CDD1D, nwl * 4 * nlyr).reshape(nwl, 4, nlyr)
    ext = np.exp(-9.0 + 10.0 * r[:, 0])
    ext *= np.minimum(1.0, 25.0 / ext.sum(axis=1, keepdims=True))
    lw = np.linspace(1.0, 0.2, nlyr)[None, :] * r[:, 3]                 # Rayleigh's share of the extinction, larger aloft
    dtaur = ext * lw
    part = ext - dtaur
    low = (np.arange(nlyr) >= (2 * nlyr) // 3)[None, :]                 # aerosol in the lowest third
    dtaua = np.where(low, 0.2 * part, 0.0)
    dtauc = part - dtaua
    wcld = 0.999999 * r[:, 1]
    waer = np.full_like(dtaua, 0.9)
    g_c = 0.9 * r[:, 2]
    g_a = 0.5 + 0.3 * r[:, 2][:, ::-1]
    tw = dtauc * wcld
    tsc = tw + dtaua * waer + dtaur                                      # depthscl's numerator, normom's dtsct
    one = np.ones_like(tw)
    lay = np.stack([dtauc, dtaua, dtaur, tsc, g_c, tw, one, g_a, dtaua, waer], axis=1)
    first = np.concatenate([[0], np.nonzero(np.diff(base.wl_of))[0] + 1])
    kidx = np.arange(W) - first[base.wl_of]
    rg = splitmix64(seed ^ 0x9E3779B9, W * nlyr).reshape(W, nlyr)
    dtaug = np.exp(-9.0 + 10.0 * rg) * (1.0 + 4.0 * kidx)[:, None]
    dtaug *= np.minimum(1.0, 25.0 / dtaug.sum(axis=1, keepdims=True))
    return MixSweep(nlyr=nlyr, nstr=nstr, nmom=base.nmom, nwl=nwl, point_of=base.wl_of.astype(np.int32), weight=base.weight,
                    dtaug=np.ascontiguousarray(dtaug), lay=np.ascontiguousarray(lay), family=(3, 3),
                    wvnmlo=base.wvnmlo[first], wvnmhi=base.wvnmhi[first], fbeam=np.ones(nwl), albedo=np.full(nwl, albedo),
                    plank=base.plank[first], temper=base.temper, umu0=base.umu0, btemp=base.btemp, ttemp=base.ttemp,
                    temis=base.temis)
