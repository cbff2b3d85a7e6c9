/*
 * sbdart_amd -- MI355X-native batched DISORT engine for SBDART's wavelength loop.
 *
 * C ABI (plain pointers and sizes; no torch / HIP types in the signatures).
 * This is the drop-in boundary for the reference's hot path:
 *
 *   reference interface being replaced                      entry point here
 *   ------------------------------------------------------  ---------------------------
 *   CALL DISORT(NLYR,DTAUC,SSALB,...,RFLDIR,RFLDN,FLUP,     sbd_engine_solve_host /
 *        DFDT,UAVG,UU,...)   drt.f:541-546, disort.f:1-6    sbd_engine_solve_device
 *     once per (wavelength, k-term) of wl_loop/kd_loop      (one call = a whole batch of
 *     (drt.f:425-561)                                        (wavelength,k) work items)
 *   per-run DISORT arguments that never change inside the   sbd_engine_create(sbd_run_cfg)
 *     loop: NSTR, TEMPER, UMU0/PHI0, UMU, PHI, BTEMP, TTEMP,
 *     TEMIS, LAMBER, ONLYFL, USRANG  (drt.f:330-335,391-421)
 *   NSTR<0 "retry with another stream count" return         SBD_ST_RETRY_NSTR status bit /
 *     (disort.f:2645-2650, drt.f:536-555)                    SBD_E_RETRY_NSTR from create
 *   errmsg warnings 2,3,4 / fatal STOPs (disutil.f:278)     per-work-item status bits
 *   stdout1's weighted spectral sums (drt.f:964-1087)       sbd_engine_accumulate_host/_device
 *   the same loop spread over the GPUs of a node: shards    sbd_fleet_create / sbd_fleet_solve_host
 *     of independent (wavelength, k) items, one sum of the   (one engine per device, one RCCL
 *     accumulators (outblk, drt.f:1-18) at the end           reduce of the accumulator block)
 *
 * The Fortran-2003 host binds these through ISO_C_BINDING
 * (sbdart_amd/fortran/sbd_engine_mod.f90); INTEGRATION.md shows the stub a
 * maintainer of the reference adds around drt.f:529-560.
 *
 * All arithmetic is fp64.  All arrays are dense, row-major by work item
 * ("[nwork][...]"), top-down in the vertical exactly like DISORT's arguments.
 * No global state: engines are independent of one another.  ONE engine drives ONE GPU and owns
 * one workspace, one stream and one staging area: use it from one thread at a time, and when
 * sbd_engine_solve_device is given a caller's stream, order that stream against the engine's
 * other uses yourself (the workspace is shared between consecutive calls).  Several GPUs of a
 * node from one process: sbd_fleet_* below.
 */
#ifndef SBDART_AMD_H
#define SBDART_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBD_ABI_VERSION 7

/* limits of the reference (params.f:9-15) */
#define SBD_MAX_NLYR 65   /* mxly   */
#define SBD_MAX_NSTR 40   /* nstrms */

/* ---- return codes (never exit(), never abort()) ---- */
#define SBD_OK               0
#define SBD_E_INVALID       -1  /* bad argument / CHEKIN-type fatal in the run config */
#define SBD_E_RETRY_NSTR    -2  /* beam angle == a quadrature angle: pick NSTR-2 / NSTR+2
                                   (disort.f:2645-2650; drt.f:536-555 does the retry).  The
                                   engine IS created (*out valid, destroy it): work items
                                   without a beam (FBEAM = 0) can still be solved with it */
#define SBD_E_NO_DEVICE     -3
#define SBD_E_HIP           -4  /* a HIP runtime call failed; see sbd_last_error() */
#define SBD_E_UNSUPPORTED   -5  /* a combination the reference itself cannot run (IBCND = 1 with USRANG and ONLYFL) */
#define SBD_E_NOMEM         -6

/* ---- per-work-item status bits ---- */
/* warnings 2/3/4: the reference raises them from LINPACK's condition estimate, 1 + RCOND == 1 (SGBCO for the band system,
 * disort.f:3607-3610; SGECO for the two dense systems, disort.f:4225-4228, 4331-4334).
 *   3, 4 (dense): LINPACK's OWN estimate.  The fast layer kernel's pivot ratio is only a filter (min/max <= 1e-10, or an
 *      eigenvalue within 1e-10 of 1/umu0, or SSALB within 1 024 ulps of 1 with a thermal source); a layer it flags goes to
 *      the reference-algorithm layer kernel, which forms the reference's matrices with one rounding per operation, factors
 *      them by SGEFA's rule and runs SGECO's estimate statement for statement (rcond_group, csrc/sbd_layer.hpp): equal status
 *      words for every ulp offset (tests/test_gpu_parity.py::test_near_singular_systems_raise_the_reference_warnings,
 *      ::test_eigenvalue_next_to_the_beam -- the latter on records for which the REFERENCE wrote SBDART_WARNING.03,
 *      tests/golden/illcond/beam_at_eigenvalue.*).
 *   2 (band): the band kernels' pivot ratio is a filter as well (1 + min|pivot|/max|pivot| == 1 or <= SBD_RCOND_FILTER); a
 *      system it flags is re-run in the reference's own formulation -- ASYMTX's eigenvectors in ASYMTX's column order
 *      (reference-algorithm layer kernel), SETMTX's matrix in LINPACK band storage, SGBFA by ISAMAX's rule and SGBCO's
 *      estimate statement for statement (band_rcond_kernel, csrc/sbd_bandco.hpp) -- and the bit is set iff 1 + RCOND == 1,
 *      a system full of NaN raising nothing, like the reference's test. */
#define SBD_ST_WARN_SOLVE0   0x01  /* band matrix singular pivot        (errmsg 2, disort.f:3609) */
#define SBD_ST_WARN_UPBEAM   0x02  /* beam-source system singular pivot (errmsg 3, disort.f:4227) */
#define SBD_ST_WARN_UPISOT   0x04  /* thermal-source system singular    (errmsg 4, disort.f:4333) */
#define SBD_ST_ERR_EIGEN     0x08  /* eigen-solve did not converge (fatal, disort.f:3254-3261) */
#define SBD_ST_RETRY_NSTR    0x10  /* disort.f:2645-2650 */
#define SBD_ST_ERR_INPUT     0x20  /* CHEKIN fatal for this item (disort.f:5140) */
#define SBD_ST_WARN_PLKAVG   0x40  /* errmsg 10: PLKAVG returns zero, possible underflow (disort.f:5657) */
#define SBD_ST_WARN_PLKCONV  0x80  /* errmsg 9: PLKAVG's Simpson rule did not converge (disort.f:5597); until round 5 it shared 0x40 */

/* flux components per output level, in this order (DISORT's output arguments) */
#define SBD_NFLUX 5
enum { SBD_RFLDIR = 0, SBD_RFLDN = 1, SBD_FLUP = 2, SBD_DFDT = 3, SBD_UAVG = 4 };

typedef struct sbd_engine sbd_engine;

/* Per-run constants: everything drt.f fixes before wl_loop starts. */
typedef struct {
    int32_t abi_version;   /* SBD_ABI_VERSION */
    int32_t nlyr;          /* NLYR = nz (drt.f:144; 1..65) */
    int32_t nstr;          /* NSTR, even, 4..40 */
    int32_t nmom;          /* highest Legendre moment supplied; PMOM row stride = nmom+1;
                              drt.f:490-494 uses min(nstr+2, 40) */
    int32_t onlyfl;        /* ONLYFL: 1 = fluxes only (iout not in 5,6,20..23) */
    int32_t lamber;        /* LAMBER: 1 = Lambertian surface with the work item's ALBEDO; 0 = bidirectional surface
                              `ibdrf` (SURFAC's quadrature of BDREF, disort.f:3765-3912; spectra.f:249-296) */
    int32_t usrang;        /* USRANG: radiances at umu[] (required when onlyfl=0) */
    int32_t numu;          /* number of user polar angles (radiance mode) */
    int32_t nphi;          /* number of user azimuths     (radiance mode) */
    int32_t nlevel_out;    /* 0 = all nlyr+1 levels (DISORT's NTAU with USRTAU=F);
                              else number of entries of level_out */
    int32_t device;        /* HIP device ordinal */
    int32_t max_batch;     /* largest nwork of one solve call (workspace is sized for
                              min(max_batch, chunk)); 0 = default */
    int32_t corint;        /* CORINT: Nakajima/Tanaka intensity corrections (INTCOR, disort.f:2044-2297) on the
                              radiances of every item with a beam and scattering (disort.f:2695-2696);
                              SBDART then supplies nmom = 299 moments (drt.f:490-491); ignored when onlyfl */
    int32_t ibdrf;         /* with lamber = 0: 1 ocean (seabdrf, isalb 7), 2 Hapke (isalb 8), 3 Ross-thick / Li-sparse
                              (isalb 9); parameters in bpar; 0 with lamber = 1 */
    double umu0;           /* cosine of solar zenith (amu0, drt.f:421,456-459) */
    double phi0;           /* solar azimuth, degrees */
    double fisot;          /* isotropic top illumination (0 in SBDART) */
    double btemp, ttemp, temis;   /* bottom/top temperature, top emissivity */
    const double *temper;  /* [nlyr+1] level temperatures, top-down (drt.f:330-333) */
    const double *umu;     /* [numu] ascending cosines (drt.f:393-403) or NULL */
    const double *phi;     /* [nphi] degrees or NULL */
    const int32_t *level_out; /* [nlevel_out] 0-based level indices (0 = TOA, nlyr = surface) */
    double bpar[8];        /* surface model parameters (albblk, spectra.f:15-26; suralb, spectra.f:61-177):
                              ocean   : wind speed (m/s), foam cover 2.951e-6 wndspd^3.52, foam reflectance 0.22 x cover
                                        (seabdrf, spectra.f:441-451), pigment concentration, salinity
                              Hapke   : single-scattering albedo, asymmetry, hot-spot amplitude, hot-spot width
                              Ross-Li : isotropic, volumetric, geometric coefficients, hot-spot magnitude, width */
    int32_t ibcnd;         /* IBCND: 0 the general case; 1 = albedo and transmissivity of the whole medium for beam
                              incidence at the output angles instead of fluxes and intensities (ALBTRN,
                              disort.f:6718-7432): umu[] then holds POSITIVE cosines (usrang = 1, onlyfl = 0 -- with
                              both set the reference overruns its UMU array), or usrang = 0 for the nstr/2 quadrature
                              cosines; results in sbd_batch_out::albtrn, ALBEDO from the work items, no sources */
    int32_t pivot_exact;   /* (a formerly reserved word: 0 keeps every caller's behaviour) NSTR <= 16: 1 = the band LU searches its pivot
                              with LINPACK's rule exactly -- ISAMAX's first maximum of |a| (disutil.f:2060-2072) -- instead of on the
                              leading 27 bits (threshold 1 - 2^-15); ~10 % more band-kernel instructions.  NSTR > 16 always does.
                              SBD_EXACT_PIVOT=1 in the environment sets it for every engine. */
} sbd_run_cfg;

/* One batch of (wavelength, k-term) work items: the per-call DISORT arguments. */
typedef struct {
    int32_t nwork;
    const double *dtauc;    /* [nwork][nlyr]            DTAUC (dtaus, drt.f:531)  */
    const double *ssalb;    /* [nwork][nlyr]            SSALB (wreal)             */
    const double *pmom;     /* [nwork][nlyr][nmom+1]    PMOM(0:nmom, lc)          */
    const double *wvnmlo;   /* [nwork]                  WVNMLO                    */
    const double *wvnmhi;   /* [nwork]                  WVNMHI                    */
    const double *fbeam;    /* [nwork]                  FBEAM (flxin, drt.f:448)  */
    const double *albedo;   /* [nwork]                  ALBEDO (rsfc, drt.f:469)  */
    const uint8_t *plank;   /* [nwork]                  PLANK (wl>2 um, drt.f:463)*/
    const double *bitem;    /* [nwork][4] ocean surface only (ibdrf = 1), else NULL: refractive index nr, ni of
                               the water and its sub-surface reflectance rsw at the item's wavelength (indwat,
                               morcasiwat: spectra.f:446-449), one spare */
    const int32_t *pmom_row;/* NULL: pmom holds nwork blocks, one per item.  Else [nwork] row indices into pmom, which then
                               holds npmom blocks [npmom][nlyr][nmom+1]: the k-terms of a spectral point share their
                               phase-function moments (drt.f:476-533 computes them once per wavelength), so the host
                               hands every block over once -- 2.3 x fewer input bytes at nk = 2.67.  Host entry points
                               want the rows non-decreasing (items in wavelength order). */
    int32_t npmom;          /* number of blocks in pmom when pmom_row != NULL */
} sbd_batch_in;

typedef struct {
    double *flux;      /* [nwork][SBD_NFLUX][nlev]  nlev = nlevel_out or nlyr+1 */
    double *uu;        /* [nwork][nphi][nlev][numu] or NULL when onlyfl */
    int32_t *status;   /* [nwork] SBD_ST_* bits */
    double *albtrn;    /* ibcnd = 1 only (else NULL / ignored): [nwork][2][nout] ALBMED then TRNMED at the nout = numu
                          (usrang) or nstr/2 output cosines; flux comes back zero like DISORT's (ZEROAL) */
} sbd_batch_out;

/* The same batch in COMPACT form (ABI v6; lay_token ABI v7): per SPECTRAL POINT what scatters there, per WORK ITEM only the gas of its
 * k-term -- the operands of the statements with which the reference turns its band model's output into DISORT's
 * arguments, which the engine then executes on the device (assemble_kernel) instead of receiving their results over
 * PCIe.  The statements, in the reference's own association (one rounding per operation, no contraction):
 *
 *   DTAUC(l) = ((dtaug(item,l) + dtauc(point,l)) + dtaua(point,l)) + dtaur(point,l)     depthscl, taugas.f:7598
 *   SSALB(l) = tsc(point,l) / DTAUC(l) where DTAUC(l) > tiny(1.d0), else 0               depthscl, taugas.f:7599-7603
 *       tsc = dtauc*wcld + dtaua*waer + dtaur, formed by the host (it is depthscl's numerator and normom's dtsct)
 *   PMOM(k,l) = [ sum over the layer's scattering terms t, in slot order, of (P_t(k) * m1_t) * m2_t
 *                 + (k == 2) 0.1 * dtaur ] / tsc   (not divided where tsc == 0);  PMOM(0,l) = 1     normom, drt.f:1390-1395
 *       P_t(k): GETMOM's moment k of the slot's phase-function family (disutil.f:2104-2209) -- 1 isotropic (0 for k >= 1),
 *       2 Rayleigh (0.1 at k = 2), 3 Henyey-Greenstein g**k (an INTEGER power: square-and-multiply from the low bit, as
 *       the reference's compiler forms it), and the Rayleigh 0.1 the reference's REAL*4 literal.
 *       A cloud layer is one term (g, taucld*wcld, 1): taucloud's TAUCLD*WCLD*PMOM/ICNT (taucloud.f:103, 132 with one
 *       cloud per layer; usrcloud taucloud.f:262-266); the boundary-layer aerosol (g, dtaua, waer): tauaero's PM*DTAUA*WAER (tauaero.f:1300);
 *       a stratospheric layer (g, dt, wa) (tauaero.f:1330).  A slot a layer does not use carries m1 = 0.
 *
 * The moments are formed once per spectral point and shared by its k-terms (as sbd_batch_in::pmom_row).  Bytes over
 * PCIe: 8 [W L + P ((4 + 3 nterm) L + 4)] against 8 W L (nmom + 3) -- a ninth at NSTR 16 with nk = 2.67.
 * Host entry points only (on the device side the assembled arrays ARE sbd_batch_in).  point_of non-decreasing; a call
 * may reference any contiguous range of the npoint blocks (only the blocks its items point at are staged).
 * Tabulated phase functions (GETMOM 4, 5; aerosol.dat / user moments), several clouds in one layer and the ocean
 * surface's per-item constants keep the arrays form (sbd_batch_in). */
#define SBD_MIX_MAX_TERMS 6
typedef struct {
    int32_t nwork, npoint;
    const int32_t *point_of;  /* [nwork]        spectral point of each work item, 0-based, non-decreasing           */
    const double *dtaug;      /* [nwork][nlyr]  absorption optical depth of the item's k-term (gases, dtaug)       */
    int32_t nterm;            /* scattering terms per layer besides Rayleigh, 0..SBD_MIX_MAX_TERMS                  */
    int32_t family[SBD_MIX_MAX_TERMS];   /* GETMOM's iphas of slot t: 1, 2 or 3                                     */
    const double *lay;        /* [npoint][4 + 3 nterm][nlyr]: channels dtauc, dtaua, dtaur, tsc, then per slot t its
                                 g, m1, m2 -- one block per spectral point, a channel's layers contiguous          */
    const double *wvnmlo, *wvnmhi, *fbeam, *albedo;   /* [npoint] as in sbd_batch_in, per spectral point            */
    const uint8_t *plank;     /* [npoint]                                                                           */
    const int32_t *kterm;     /* NULL, or with dtaug == NULL: [nwork] k-term (0-based) of each item -- its gas depth is
                                 the one sbd_fleet_gas_terms left ON THE DEVICE for (point_of, kterm): the gas never
                                 crosses PCIe.  point_of then counts the points of that gas call, and a fleet of several
                                 devices hands every item to the device that holds its point */
    int64_t lay_token;        /* (ABI v7) 0: the layer blocks are staged from `lay`.  Non-zero: the generation number
                                 sbd_fleet_gas_terms / sbd_fleet_point_terms returned for the blocks it left (or made) ON
                                 THE DEVICES (with dtaug == NULL only): the solve reads those, `lay` is not touched and
                                 may be NULL; a number that is not the fleet's
                                 current one is SBD_E_INVALID.  Residency is the caller's explicit statement, never
                                 inferred from pointer values */
} sbd_mix_in;

/* The gas part of the band model for a run (ABI v6): LOWTRAN7's band model and continua along a vertical and a slant
 * path, the three-term k-distribution fit, the Newton slant-path correction and depthscl's KDIST policy (gasset / taugas
 * / kdistr / taucor / depthscl, taugas.f:7392-7510, 2236-2534 with the look-ups 2538-6821, 1802-1920, 7650-7692,
 * 7550-7590) -- what the reference evaluates per wavelength between its table look-ups and DISORT's optical depths,
 * evaluated on the device for all wavelengths of a run at once (north_star: "per-wavelength optical depths ...
 * precomputed into coalesced HBM arrays").  The same sequence of roundings as the reference; exp / log / log10 / pow are
 * the device math library's (a few ulps per call: tests/test_gas_device.py states the bound on the optical depths). */
#define SBD_GAS_SLOTS 63   /* absorber-amount slots, params.f:14 (mxq) */
typedef struct {
    int32_t nz;            /* levels = DISORT layers (drt.f:144; with SPOWDER's extra level)                        */
    int32_t kdist;         /* KDIST 0..3 (taugas.f:7520-7528)                                                       */
    const double *uu;      /* [nz][SBD_GAS_SLOTS] absorber amounts above each level, levels bottom-up (absint,
                              taugas.f:1924-2233)                                                                    */
    const double *z;       /* [nz] level altitudes, km, bottom-up                                                  */
    double amu0_first;     /* cosine of the solar zenith angle for the gas terms of the run's FIRST wavelength ...  */
    double amu0_rest;      /* ... and of every later one (the same unless SZA >= 90: drt.f:433-455)                 */
    double xo4;            /* XO4 (taugas.f:3190)                                                                   */
    const void *tables;    /* image of sbdart_amd/data/sbdart_tables.bin (the band model's coefficient tables)      */
    size_t tables_bytes;
} sbd_gas_model;

/* The scatterers' part of the band model for a run (ABI v7, round 6): Rayleigh depths, the cloud deck (Mie look-up) and the
 * boundary-layer / stratospheric aerosols (rayleigh spectra.f:179-247; taucloud.f:10-140, 6726-6768; tauaero.f:1175-1359) --
 * what the reference evaluates per wavelength to fill the operands sbd_mix_in::lay holds, evaluated on the device for all
 * wavelengths of a run at once (sbd_fleet_point_terms): the layer blocks are BORN in HBM and never cross PCIe.  Covered: what
 * the compact form covers (one cloud per layer from ZCLOUD / TCLOUD / LWP / NRE with IMOMC 1..3; IAER 1..5 with IMOMA 1..3;
 * stratospheric layers JAER / TAERST); usrcld.dat, aerosol.dat, user moments and SPOWDER keep the host's blocks.  The same
 * sequence of roundings as the reference (csrc/sbd_scat.hpp, pinned bit for bit on the host by sbd_scatter_blocks_host). */
#define SBD_SCAT_SLOTS 5   /* ncldz = naerz = 5 (params.f:12) */
typedef struct {
    int32_t nz;                  /* levels = layers (drt.f:144)                                                          */
    const double *z, *p, *t;     /* [nz] level altitude (km), pressure (mb), temperature (K), bottom-up (atms)           */
    double xrsc;                 /* XRSC: multiplier of the Rayleigh depths (drt.f:470)                                  */
    int32_t cloud_term;          /* 1: the run has a cloud deck -- it is the FIRST scattering term of every block        */
    int32_t cld_nslot;           /* cloud slots in use; per slot: layer (1 = top; negative: "the cloud extends up to"),  */
    int32_t cld_layer[SBD_SCAT_SLOTS];                     /* TCLOUD, LWP, NRE (taucloud.f:40-100)                        */
    double cld_tcloud[SBD_SCAT_SLOTS], cld_lwp[SBD_SCAT_SLOTS], cld_nre[SBD_SCAT_SLOTS];
    int32_t iaer, nosct;         /* IAER (0 none, else the boundary-layer spectrum below), NOSCT                         */
    int32_t aer_nwl;             /* boundary-layer spectrum: wavelengths, extinction, absorption, asymmetry factor       */
    const double *aer_wl, *aer_ext, *aer_absb, *aer_asym;   /* [aer_nwl] (tauaero.f:1240-1290)                            */
    double abaer;                /* ABAER: Angstrom exponent outside the spectrum                                         */
    const double *aer_column;    /* [nz] layers top-down: boundary-layer depth at 0.55 um / extinction(0.55)             */
    int32_t nstrat;              /* stratospheric layers: model JAER, layer (1 = top), depth at 0.55 um TAERST           */
    int32_t jaer[SBD_SCAT_SLOTS], strat_layer[SBD_SCAT_SLOTS];
    double taerst[SBD_SCAT_SLOTS];
    const void *tables;          /* image of sbdart_amd/data/sbdart_tables.bin (Mie and stratospheric tables)            */
    size_t tables_bytes;
} sbd_scat_model;

/* ---- lifecycle ---- */
int  sbd_engine_create(const sbd_run_cfg *cfg, sbd_engine **out);
void sbd_engine_destroy(sbd_engine *e);

/* ---- the hot path ---- */
/* Pointers in `in`/`out` are DEVICE pointers (HBM-resident, 8-byte aligned).
 * `hip_stream` is a hipStream_t passed as void* (NULL = the engine's own stream).
 * Asynchronous: returns after enqueueing; results are ordered on the stream. */
int sbd_engine_solve_device(sbd_engine *e, const sbd_batch_in *in, const sbd_batch_out *out,
                            void *hip_stream);
/* Pointers are HOST pointers: stages H2D, solves, copies back, synchronises. */
int sbd_engine_solve_host(sbd_engine *e, const sbd_batch_in *in, const sbd_batch_out *out);

/* stdout1's reduction (drt.f:964-1054): acc[c][lev] += sum_i weight[i] * flux[i][c][lev],
 * and acc_uu[phi][lev][mu] += sum_i weight[i]*uu[i][...] when uu != NULL.
 * fp64, deterministic order: 256-item segments summed in work-item order, then the segment
 * sums added in order (independent of the launch shape and of the GPU). */
int sbd_engine_accumulate_device(sbd_engine *e, int32_t nwork, const double *weight,
                                 const double *flux, const double *uu,
                                 double *acc_flux, double *acc_uu, void *hip_stream);
int sbd_engine_accumulate_host(sbd_engine *e, int32_t nwork, const double *weight,
                               const double *flux, const double *uu,
                               double *acc_flux, double *acc_uu);

/* ---- several GPUs, one process (the reference's loop has no cross-item dependence, drt.f:425-561) ----
 * A fleet = one engine per device (devices == NULL or ndev <= 0: every visible device).  A batch is
 * cut into contiguous shards, sbd_shard_range(nwork, nshard, rank): the first nwork % nshard shards hold
 * one item more.  sbd_fleet_solve_host: HOST pointers; every device stages and solves its shard on its
 * own stream; per-item outputs land in place in `out` (out->flux / out->uu may be NULL when only the
 * sums are wanted, out->status is required).  With weight != NULL the weighted sums of the whole batch
 * (stdout1's accumulation, same element order as sbd_engine_accumulate_*) are ADDED to acc_flux
 * [SBD_NFLUX][nlev] and, if not NULL, acc_uu [nphi][nlev][numu]: per-device fixed-order sums, then one
 * ncclReduce(sum, double) onto the first device (RCCL over xGMI) -- or, when a device appears twice in
 * the list or RCCL is unavailable, a host-side sum in device order.
 * sbd_fleet_create returns SBD_E_RETRY_NSTR like sbd_engine_create (the fleet is created). */
typedef struct sbd_fleet sbd_fleet;
int      sbd_fleet_create(const sbd_run_cfg *cfg, int32_t ndev, const int32_t *devices, sbd_fleet **out);
void     sbd_fleet_destroy(sbd_fleet *f);
int32_t  sbd_fleet_size(const sbd_fleet *f);
sbd_engine *sbd_fleet_engine(sbd_fleet *f, int32_t i);      /* the i-th device's engine (introspection) */
int32_t  sbd_fleet_uses_rccl(const sbd_fleet *f);           /* 1: sums are reduced by RCCL, 0: on the host */
void     sbd_shard_range(int32_t nwork, int32_t nshard, int32_t rank, int32_t *lo, int32_t *hi);
int      sbd_fleet_solve_host(sbd_fleet *f, const sbd_batch_in *in, const sbd_batch_out *out,
                              const double *weight, double *acc_flux, double *acc_uu);
/* ... from the compact form (sbd_mix_in, HOST pointers): every pass stages its slice of the compact arrays and
 * assembles DTAUC / SSALB / PMOM on the device ahead of its kernels.  Several devices: the batch is cut BETWEEN
 * spectral points -- sbd_shard_range_points: the item boundaries of sbd_shard_range, each moved up to the next item
 * that starts a spectral point (the k-terms of a point stay on one device, which forms the point's moments once) --
 * and summed like sbd_fleet_solve_host.  Lambertian surface or a bidirectional one without per-item constants
 * (ibdrf 0, 2, 3). */
void     sbd_shard_range_points(int32_t nwork, const int32_t *point_of, int32_t nshard, int32_t rank, int32_t *lo, int32_t *hi);
int      sbd_fleet_solve_mix_host(sbd_fleet *f, const sbd_mix_in *in, const sbd_batch_out *out,
                                  const double *weight, double *acc_flux, double *acc_uu);
/* The gas terms of npoint wavelengths on the fleet's devices (points sharded by sbd_shard_range, every device keeps its
 * points' results): wl [npoint] micrometres (wl[0] is the run's first wavelength), lay [npoint][nch][nlyr] the points'
 * layer blocks in sbd_mix_in's layout -- the gas model reads channels 0..2 only (dtauc, dtaua, dtaur: depthscl's
 * roll-off of the k-terms looks at the scatterers' depth above a level, taugas.f:7550-7590), so nch >= 3; a caller that
 * will solve from the same blocks passes nch = 4 + 3 nterm.  Returns per point nk (1 or 3 k-terms), wt [npoint][3] the
 * terms' weights (depthscl's wt: 1 when nk = 1), and taucor's failures: fail [npoint] 1 where the reference would stop
 * ("TAUCOR: iteration did not converge"), may be NULL.  The depths stay on the devices for sbd_fleet_solve_mix_host with
 * dtaug == NULL.  So do the layer blocks, and the call says so explicitly: *lay_token (may be NULL) receives a non-zero
 * generation number naming the device copy of `lay`; a later sbd_mix_in with lay_token set to that number solves from the
 * resident blocks (its `lay` pointer is not read), with lay_token = 0 the blocks are staged from `lay` again.  A token is
 * valid until the next sbd_fleet_gas_terms call on the fleet; a stale one is SBD_E_INVALID, never silently stale data.
 * dtaug_out, if not NULL: [npoint][3][nlyr] the terms' gas depths copied back (tests, IOUT-independent inspection). */
int      sbd_fleet_gas_terms(sbd_fleet *f, const sbd_gas_model *g, int32_t npoint, const double *wl, const double *lay,
                             int32_t nch, int32_t *nk, double *wt, int32_t *fail, double *dtaug_out, int64_t *lay_token);
/* sbd_fleet_gas_terms with the layer blocks made ON THE DEVICES from the scatterers' model (sbd_scat_model above) instead
 * of coming from the host: per point of `wl` the block [nch][nlyr] (nch = 4 + 3 x the model's terms: cloud, boundary-layer
 * aerosol, one per active stratospheric layer -- sbd_mix_in's layout) is computed where the gas kernel and the solves read
 * it.  Everything else as sbd_fleet_gas_terms; lay_out, if not NULL: [npoint][nch][nlyr] the blocks copied back (tests, and
 * the host's report on an item CHEKIN refuses). */
int      sbd_fleet_point_terms(sbd_fleet *f, const sbd_gas_model *g, const sbd_scat_model *sm, int32_t npoint, const double *wl,
                               int32_t nch, int32_t *nk, double *wt, int32_t *fail, double *dtaug_out, double *lay_out,
                               int64_t *lay_token);
/* The scatterers' blocks on the HOST (no GPU; the same source, csrc/sbd_scat.hpp): bit-equal to the Fortran host's band model,
 * hence to the reference's operands -- the pin of the device kernel's source.  lay_out [npoint][nch][nz]. */
int      sbd_scatter_blocks_host(const sbd_scat_model *sm, int32_t npoint, const double *wl, int32_t nch, double *lay_out);
/* The same arithmetic on the HOST (no GPU involved; the same source, sbd_gas.hpp): with the host's libm the results
 * are bit-equal to the Fortran host's band model, hence to the reference's -- the pin of the device kernel's source. */
int      sbd_gas_terms_host(const sbd_gas_model *g, int32_t nlyr, int32_t npoint, const double *wl, const double *lay, int32_t nch,
                            int32_t *nk, double *wt, int32_t *fail, double *dtaug_out);

/* errmsg 2 on the HOST (no GPU involved): LINPACK's reciprocal condition estimate (SGBCO, disutil.f:426-769) of the
 * boundary-value system of azimuth mode `mazim` of ONE work item over a Lambertian surface, as SOLVE0 forms and tests it
 * (disort.f:3602-3610) -- the reference's own formulation restated (csrc/sbd_refband.hpp: SETDIS's scaling, SOLEIG on ASYMTX,
 * SETMTX in LINPACK band storage, SGBFA / SGBCO), the SAME source band_rcond_kernel runs on the device for the systems the
 * band kernels' filter flags.  With the host's exp the value is the reference's bit for bit (tests/test_refband_host.py
 * against the oracle, which is pinned on the reference executable's SBDART_WARNING.02 files): the pin of that kernel's
 * source.  dtauc / ssalb [nlyr], pmom [nlyr][nmom+1] as in sbd_batch_in; plank only decides LYRCUT (disort.f:2602).
 * SBD_E_UNSUPPORTED when ASYMTX does not converge (the reference stops there). */
int      sbd_band_rcond_host(int32_t nlyr, int32_t nstr, int32_t nmom, int32_t mazim, int32_t plank, double albedo,
                             const double *dtauc, const double *ssalb, const double *pmom, double *rcond);

/* How the devices are fed (replaces nothing in the reference: its loop is serial, drt.f:425-561): every device's
 * shard is enqueued from a host thread of its own, and when the fleet spans several devices (or SBD_PIN_INPUTS=1)
 * dtauc / ssalb / pmom are page-locked for the duration of the call, so pageable arrays of the caller do not
 * serialise the devices.  sbd_fleet_last_enqueue: host clock (seconds since the last call began) around device
 * i's enqueue and the number of arrays that call page-locked.  sbd_host_alloc / sbd_host_free: page-locked
 * host memory for a host program's batch arrays (then nothing is registered per call).
 * SBD_FLEET_RCCL=1 in the environment makes a fleet of ONE device reduce through RCCL too (nranks = 1). */
int      sbd_fleet_last_enqueue(const sbd_fleet *f, int32_t i, double *t_begin, double *t_end, int32_t *npinned);
int      sbd_host_alloc(size_t bytes, void **out);
void     sbd_host_free(void *p);

/* DREF (disort.f:5178-5284) on the host: flux albedo of the bidirectional surface model `ibdrf` (1 ocean, 2 Hapke,
 * 3 Ross-Li; bpar / bitem as in sbd_run_cfg / sbd_batch_in, bitem = NULL for 2 and 3) for incidence cosine mu,
 * |mu| <= 1 as in the reference (its driver passes cos(SZA) also when the sun is below the horizon).  What
 * drt.f:478-484 calls per wavelength for ISALB -7, -8, -9 (a Lambertian surface with that albedo, clamped to [0,1]
 * by the caller; the reference warns when the value lies outside) and what CHEKIN's test of a bidirectional surface
 * prints (disort.f:5080-5096).  No GPU involved. */
int      sbd_surface_flux_albedo(int32_t ibdrf, const double *bpar, const double *bitem, double mu, double *albedo);

/* ---- introspection ---- */
int32_t     sbd_abi_version(void);
int32_t     sbd_engine_nlevel(const sbd_engine *e);     /* nlev of the outputs */
size_t      sbd_engine_workspace_bytes(const sbd_engine *e);
int32_t     sbd_engine_chunk(const sbd_engine *e);      /* work items per internal pass */
int32_t     sbd_engine_pass_count(const sbd_engine *e, int32_t nwork);   /* equal passes a resident batch of nwork items is cut into */
void       *sbd_engine_stream(sbd_engine *e);           /* the engine's hipStream_t */
/* Gauss quadrature the engine uses (QGAUSN, disort.f:5984): cmu/cwt get nstr/2 values */
int         sbd_engine_quadrature(const sbd_engine *e, double *cmu, double *cwt);
/* wall time (ms) of the kernels of the most recent solve_device call, measured with HIP
 * events on the stream the kernels ran on; phase: 0 setup, 1 layer, 2 band LU,
 * 3 back-substitution + fluxes, 4 intensities, -1 total.  Synchronises the stream. */
double      sbd_engine_last_ms(sbd_engine *e, int phase);
/* on = 1: ONE stream, a synchronisation per pass -- each kernel family alone on the chip; on = 2: the events are recorded
 * where the passes run (two streams, a pass's kernels beside the other pass's) and read at the next sbd_engine_last_ms:
 * the durations a kernel trace of the same call shows; 0: off */
void        sbd_engine_enable_timing(sbd_engine *e, int on);
/* timing mode: number of (item, mode, layer) eigenproblems of the last solve that the fast layer kernel handed
 * to the reference-algorithm kernel (not positive definite after symmetrisation, no Jacobi convergence, thermal
 * source in a conservative layer); -1 without timing */
int64_t     sbd_engine_last_fallback_layers(sbd_engine *e);
/* Test hook: copy one workspace array of the LAST chunk solved to the host.
 * which: 0 gc, 1 kk, 2 ek, 3 zz, 4 zp0, 5 zp1, 6 ll, 7 sv, 8 svi(int32), 16 band_rcond_kernel's estimates [item x mode]
 * (NaN where no system was served), 17 its list (int32: count, then item x mode indices).  Returns bytes copied
 * (<= nbytes) or a negative error. */
long long   sbd_engine_debug_copy(sbd_engine *e, int which, void *host_buf, size_t nbytes);
/* Test hook (NSTR <= 16, all output levels): on != 0 makes the band LU record, per system and elimination
 * sub-step, the register index of the pivot row it took (which = 15 of sbd_engine_debug_copy, int32 [item x mode]
 * [NLYR x NSTR]); the test replays the window bookkeeping and compares the ROWS with the ones SGBFA's ISAMAX takes
 * (disutil.f:852-912, 2060-2072). */
int         sbd_engine_debug_pivots(sbd_engine *e, int on);
const char *sbd_strerror(int code);
const char *sbd_last_error(void);   /* thread-local detail for SBD_E_HIP / SBD_E_INVALID */

#ifdef __cplusplus
}
#endif
#endif /* SBDART_AMD_H */
