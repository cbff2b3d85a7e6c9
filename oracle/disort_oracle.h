/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64) of the reference's DISORT solve as SBDART's
 * wavelength loop drives it (drt.f:541-546 -> disort.f:1-871).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker.  The product path is sbdart_amd/csrc (HIP).
 *
 * Parity status: PINNED.  tests/test_oracle_*.py check this restatement
 * against (a) the reference's built-in known answers (SLFTST, disort.f:6446-6449),
 * (b) DISORT input/output records captured from the reference executable built
 * from /root/reference (oracle/build_ref.sh -> the tests/golden .sbdrec files), and
 * (c) when oracle/_ref is present, the reference DISORT itself on fresh inputs.
 */
#ifndef SBD_DISORT_ORACLE_H
#define SBD_DISORT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* status bits returned in sbdo_out.status (reference: errmsg numbers) */
#define SBDO_WARN_SOLVE0_RCOND 0x01   /* errmsg(2)  disort.f:3609 */
#define SBDO_WARN_UPBEAM_RCOND 0x02   /* errmsg(3)  disort.f:4227 */
#define SBDO_WARN_UPISOT_RCOND 0x04   /* errmsg(4)  disort.f:4333 */
#define SBDO_ERR_ASYMTX        0x08   /* fatal      disort.f:3254-3261 */
#define SBDO_RETRY_NSTR        0x10   /* nstr=-abs(nstr) disort.f:2645-2650 */
#define SBDO_ERR_INPUT         0x20   /* CHEKIN fatal, disort.f:5140 */
#define SBDO_WARN_PLKAVG       0x40   /* errmsg(10) disort.f:5657 */
#define SBDO_WARN_PLKCONV      0x80   /* errmsg(9) disort.f:5597 */

typedef struct {
    int nlyr, nstr, nmom;       /* PMOM row stride is nmom+1 */
    int numu, nphi;             /* user angles (radiance mode only) */
    int plank, onlyfl, lamber, usrang;
    int usrtau, ntau;           /* usrtau=0 in SBDART; kept for SLFTST */
    double wvnmlo, wvnmhi, fbeam, umu0, phi0, fisot, albedo, btemp, ttemp, temis;
    double accur;               /* azimuth-series convergence; SBDART passes 0 (drt.f:142) */
    int corint;                 /* Nakajima/Tanaka intensity corrections (INTCOR, disort.f:2044) */
    const double *dtauc;        /* [nlyr]            top-down */
    const double *ssalb;        /* [nlyr]                     */
    const double *temper;       /* [nlyr+1]  levels 0..nlyr   */
    const double *pmom;         /* [nlyr][nmom+1]             */
    const double *umu;          /* [numu] ascending           */
    const double *phi;          /* [nphi] degrees             */
    const double *utau;         /* [ntau] if usrtau           */
    /* bidirectional surface (LAMBER off), spectra.f:249-296: 0 none, 1 ocean (seabdrf), 2 Hapke, 3 Ross-Li.
       bpar: the model's run parameters, bitem: the ocean's per-wavelength constants nr, ni, rsw
       (layout: sbdart_amd/records.py) */
    int ibdrf;
    double bpar[8], bitem[4];
    int ibcnd;                  /* 1: albedo and transmissivity of the whole medium for beam incidence at the user
                                   angles (ALBTRN, disort.f:6718-7432) instead of fluxes and intensities */
} sbdo_in;

typedef struct {
    int nstr_out;               /* -nstr when the beam angle hits a quadrature angle */
    int status;
    int ntau;
    double *rfldir, *rfldn, *flup, *dfdt, *uavg;  /* [ntau] (ntau = nlyr+1 unless usrtau) */
    double *uu;                 /* [nphi][ntau][numu], may be NULL when onlyfl */
    double *u0c;                /* optional [ntau][nstr] azimuthal-mean intensities at
                                   quadrature angles (FLUXES' U0C), may be NULL */
    /* optional debug dumps of the azimuth mode `dbg_mode` (all may be NULL):
       gc [nlyr][nstr(j)][nstr(i)] = GC(i,j,lc) column-major per layer; kk, ll, zz, zplk0,
       zplk1 [nlyr][nstr] */
    int dbg_mode;
    double *dbg_gc, *dbg_kk, *dbg_ll, *dbg_zz, *dbg_zplk0, *dbg_zplk1;
    double *albmed, *trnmed;    /* IBCND = 1: [numu] (or [nstr/2] when USRANG is off), else untouched; may be NULL */
    int *dbg_ipvt;              /* [nlyr*nstr] SGBFA's pivot rows of that mode's band system, 1-based (disutil.f:852-912) */
} sbdo_out;

int  sbdo_disort(const sbdo_in *in, sbdo_out *out);
/* smallest LINPACK condition estimate of the most recent sbdo_disort call on this thread: which = 0 band system
   (SGBCO, disort.f:3607), 1 UPBEAM's, 2 UPISOT's (SGECO, disort.f:4225, 4331); +inf when none was formed.
   A search aid for fixtures in which the reference's 1 + RCOND == 1 fires. */
double sbdo_last_rcond(int which);

/* building blocks, exported for unit tests */
void   sbdo_qgausn(int m, double *gmu, double *gwt);                 /* disort.f:5984 */
void   sbdo_lepoly(int nmu, int m, int maxmu, int twonm1,
                   const double *mu, double *ylm /* [nmu][maxmu+1] */); /* disort.f:5286 */
double sbdo_plkavg(double wnumlo, double wnumhi, double t, int *warn);  /* disort.f:5410 */
int    sbdo_asymtx(double *aa, double *evec, double *eval, int m,
                   int ia, int ievec, double *wk);                    /* disort.f:873 */
void   sbdo_sgbfa(double *abd, int lda, int n, int ml, int mu, int *ipvt, int *info);
void   sbdo_sgbsl(const double *abd, int lda, int n, int ml, int mu,
                  const int *ipvt, double *b);
double sbdo_sgbco(double *abd, int lda, int n, int ml, int mu, int *ipvt, double *z);
void   sbdo_sgefa(double *a, int lda, int n, int *ipvt, int *info);
void   sbdo_sgesl(const double *a, int lda, int n, const int *ipvt, double *b);
double sbdo_sgeco(double *a, int lda, int n, int *ipvt, double *z);

/* reference constants (fp32 literals widened to fp64) */
double sbdo_dref(int ibdrf, const double *bpar, const double *bitem, double mu);   /* disort.f:5178-5284 */
double sbdo_pi(void);      /* 2.*ASIN(1.0) in fp32 = 3.14159274101257324 (disort.f:441) */
double sbdo_dither(void);  /* 100*2^-52 (disort.f:442-448) */

#ifdef __cplusplus
}
#endif
#endif
