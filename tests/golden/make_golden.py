#!/usr/bin/env python3
"""Generate the golden DISORT records + stdout goldens under tests/golden/.

Runs ONLY in the build container (needs /root/reference and amdflang):

    ./oracle/build_ref.sh && python tests/golden/make_golden.py

For every case below it writes the INPUT namelist the reference's own test
script uses (TestRuns/test_runs:31-145) or a BASELINE.json config, runs
oracle/_ref/sbdart_capture (the unmodified reference objects with the DISORT
call site interposed, see oracle/ref/sbd_ref_capture.f90) and keeps a
sub-sample of the captured DISORT input/output records, plus the stdout.

Outputs are DATA (inputs + expected outputs): no reference source is copied.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from sbdart_amd.records import read_records, write_records  # noqa: E402

CAPTURE = os.path.join(ROOT, "oracle", "_ref", "sbdart_capture")


def run_case(namelist: str, files=None):
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "INPUT"), "w") as f:
            f.write("\n &INPUT\n" + namelist + "\n /\n")
        for name, text in (files or {}).items():        # the user's data files (atms.dat, usrcld.dat, filter.dat, ...)
            with open(os.path.join(d, name), "w") as f:
                f.write(text)
        env = dict(os.environ, SBD_CAPTURE_FILE=os.path.join(d, "cap.sbdrec"))
        out = subprocess.run([CAPTURE], cwd=d, env=env, capture_output=True, text=True, check=True).stdout
        recs = read_records(os.path.join(d, "cap.sbdrec"))
        warns = sorted(f for f in os.listdir(d) if f.startswith("SBDART_WARNING"))
    return out, recs, warns


def every_nth_wl(recs, n, offset=0):
    return [r for r in recs if (r.iwl - 1 - offset) % n == 0]


def main():
    manifest = {}

    only = set(sys.argv[1:])                      # names to (re)generate; none given = all
    if only and os.path.exists(os.path.join(HERE, "MANIFEST.json")):
        manifest.update(json.load(open(os.path.join(HERE, "MANIFEST.json"))))

    def emit(name, namelists, pick, keep_stdout=True, full_inputs=False, files=None):
        if only and name not in only:
            return None
        allrec, stdout, allw = [], "", []
        for nl in namelists:
            out, recs, warns = run_case(nl, files)
            stdout += out
            allrec += pick(recs)
            allw += warns
        path = os.path.join(HERE, name + ".sbdrec")
        write_records(path, allrec, with_out=True)
        entry = {"namelists": namelists, "records": len(allrec),
                 "bytes": os.path.getsize(path),
                 "sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(),
                 "reference_warnings": sorted(set(allw))}
        if keep_stdout:
            with open(os.path.join(HERE, name + ".stdout"), "w") as f:
                f.write(stdout)
        manifest[name] = entry
        print(name, entry["records"], "records", entry["bytes"], "bytes", entry["reference_warnings"])
        return allrec

    # --- TestRuns example 1 (test_runs:31-39): keep EVERY record (inputs+outputs): the
    #     Fortran host replays this file end to end and must reproduce sbchk.1's stdout.
    emit("sbchk1", ["    idatm=4,   isat=0, wlinf=.25, wlsup=1.0, wlinc=.005, iout=1,"],
         lambda r: r)
    # --- example 2 (test_runs:45-66): 48 single-wavelength runs, all kept
    emit("sbchk2", [f" tcloud={t}\n albcon={a}\n idatm=4\n isat=0\n wlinf=.55\n wlsup=.55\n"
                    f" isalb=0\n iout=10\n sza=30"
                    for a in ("0", ".2", ".4", ".6", ".8", "1")
                    for t in (0, 1, 2, 4, 8, 16, 32, 64)], lambda r: r)
    # --- example 3 (test_runs:72-94): thermal, every 12th wavelength
    emit("sbchk3", [f"  tcloud={t}\n  zcloud=8\n  nre=10\n  idatm=4\n  sza=95\n  wlinf=4\n"
                    f"  wlsup=20\n  wlinc=-.01\n  iout=1" for t in (0, 1, 5)],
         lambda r: every_nth_wl(r, 12))
    # --- example 4 (test_runs:99-119): every 3rd case
    cases4 = [f" tcloud={t}\n nre={n}\n wlinf={w}\n wlsup={w}\n idatm=1\n isat=0\n isalb=6\n"
              f" iout=10\n sza=0"
              for t in (0, 1, 2, 4, 8, 16, 32, 64, 128) for n in (2, 4, 8, 16, 32, 64, 128)
              for w in (".55", "2.16")]
    emit("sbchk4", cases4[::3], lambda r: r)
    # --- example 5 (test_runs:124-145): nstr=20 radiance
    emit("sbchk5", [f"  tcloud= {t}\n  zcloud= 1\n  wlinf=.72\n  wlsup=.72\n  idatm=1\n  isalb=4\n"
                    f"  sza=60\n  iout=21\n  nstr=20\n"
                    f"  uzen=5,15,25,35,45,55,65,75,85,95,105,115,125,135,145,155,165,175\n"
                    f"  phi=0,15,30,45,60,75,90,105,120,135,150,165,180" for t in (5, 15)],
         lambda r: r)
    # --- BASELINE.json configs ---
    emit("cfgA_sw_nstr4", ["idatm=6 isat=0 wlinf=.25 wlsup=4.0 wlinc=.005 iout=1 nstr=4"],
         lambda r: every_nth_wl(r, 60), keep_stdout=False)
    emit("cfgB_sw_nstr16", ["idatm=6 isat=0 wlinf=.25 wlsup=4.0 wlinc=.005 iout=1 nstr=16"],
         lambda r: every_nth_wl(r, 40, 3), keep_stdout=False)
    emit("cfg3_lw_nstr16_cloud",
         ["idatm=6 wlinf=4 wlsup=80 wlinc=-.01 nstr=16 tcloud=10 zcloud=1 nre=8 sza=95 iout=1"],
         lambda r: every_nth_wl(r, 25, 2), keep_stdout=False)
    emit("cfgC_rad_nstr32",
         ["idatm=6 wlinf=.5 wlsup=.9 wlinc=.2 iout=5 nstr=32 iaer=1 vis=23 nzen=20 uzen=0,85 "
          "nphi=16 phi=0,180 sza=30"], lambda r: r[:3], keep_stdout=False)
    emit("cfgD_nstr32_50ly",
         ["idatm=6 wlinf=.25 wlsup=100 wlinc=20 nstr=32 ngrid=50 iout=10 sza=30"],
         lambda r: every_nth_wl(r, 250, 7), keep_stdout=False)

    # --- round 5 ("full in shape, thin in extent"): BASELINE configs[3] across its whole spectral
    #     range -- one record per wavelength of a 26-point sub-grid of 0.25-4.0 um (the point's last k-term: the
    #     strongest absorption), NSTR 32, 20 x 16 angles, rural aerosol; thermal source switched on by the reference
    #     itself above 2 um --
    def last_term_of_each_point(recs):
        out = {}
        for r in recs:
            out[r.iwl] = r
        return [out[k] for k in sorted(out)]
    emit("cfgC_rad_nstr32_wide",
         ["idatm=6 wlinf=.25 wlsup=4.0 wlinc=.15 iout=5 nstr=32 iaer=1 vis=23 nzen=20 uzen=0,85 nphi=16 phi=0,180 sza=30"],
         last_term_of_each_point, keep_stdout=False)
    #     ... and configs[4] at its finest spectral step (1 cm-1) in the thermal tail, 20-25 um: every fourth point
    emit("cfgD_thermal_tail_1cm",
         ["idatm=6 wlinf=20 wlsup=25 wlinc=1.0001 nstr=32 ngrid=50 iout=10 sza=30"],
         lambda r: every_nth_wl(r, 4, 1), keep_stdout=False)
    #     ... and the conservative-thermal class (SSALB = 1 exactly in a layer with a thermal source: DISORT dithers it
    #     to 1 - 2.2e-14, I - CC is singular to ten digits and the particular solution cancels against the homogeneous
    #     one) as REFERENCE records: the reference's DISORT called directly (oracle/_ref/disort_ref_cli) on synthetic
    #     columns -- one to five layers, optical depths 0.1 to 50, NSTR 4 to 32, the conservative layer in the middle,
    #     at the top and at the bottom, with and without a beam
    if not only or "conservative_thermal" in only:
        import numpy as np
        from sbdart_amd.records import F_LAMBER, F_ONLYFL, F_PLANK, SolveRecord
        recs = []
        for nstr in (4, 8, 16, 24, 32):
            k = np.arange(nstr + 3)
            for nlyr, tau, where, fbeam in ((3, 0.1, 1, 1.0), (3, 1.0, 1, 1.0), (1, 10.0, 0, 1.0), (5, 50.0, 2, 0.0), (4, 2.0, 0, 1.0),
                                            (4, 0.5, 3, 0.0)):
                ss = np.full(nlyr, 0.5)
                ss[where] = 1.0
                recs.append(SolveRecord(nlyr=nlyr, nstr=nstr, nmom=nstr + 2, flags=F_LAMBER | F_PLANK | F_ONLYFL, wvnmlo=900.0,
                                        wvnmhi=950.0, fbeam=fbeam, umu0=0.6, phi0=0.0, albedo=0.3, btemp=300.0, ttemp=200.0, temis=0.5,
                                        dtauc=np.full(nlyr, tau), ssalb=ss, temper=np.linspace(220.0, 295.0, nlyr + 1),
                                        pmom=np.full(nlyr, 0.6)[:, None] ** k[None, :], umu=np.zeros(0), phi=np.zeros(0)))
        with tempfile.TemporaryDirectory() as d:
            write_records(os.path.join(d, "in.sbdrec"), recs, with_out=False)
            subprocess.run([os.path.join(ROOT, "oracle", "_ref", "disort_ref_cli"), "in.sbdrec", "out.sbdrec", "1"],
                           cwd=d, check=True, capture_output=True)
            out = read_records(os.path.join(d, "out.sbdrec"))
        path = os.path.join(HERE, "conservative_thermal.sbdrec")
        write_records(path, out, with_out=True)
        manifest["conservative_thermal"] = {"namelists": ["(disort_ref_cli on synthetic columns with SSALB = 1 in one layer and a thermal source)"],
                                            "records": len(out), "bytes": os.path.getsize(path),
                                            "sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(), "reference_warnings": []}
        print("conservative_thermal", len(out), "records")

    # --- intensity corrections (CORINT = true: 299 phase-function moments per layer), one record each:
    #     a cloud seen from below and above incl. the solar aureole (IMS term, viewing angles within 10
    #     degrees of the beam), rural aerosol, and a thermal + solar point over a bright surface
    emit("corint_nstr8",
         ["idatm=6 wlinf=.55 wlsup=.55 iout=20 nstr=8 corint=t tcloud=3 zcloud=2 nzen=8 uzen=0,175 nphi=3 phi=0,180 sza=40",
          "idatm=2 wlinf=.45 wlsup=.45 iout=20 nstr=8 corint=t iaer=1 vis=10 nzen=6 uzen=100,170 nphi=2 phi=0,90 sza=25 imoma=4",
          "idatm=4 wlinf=3.8 wlsup=3.8 iout=20 nstr=8 corint=t tcloud=1 zcloud=4 nre=-30 albcon=.6 nzen=5 uzen=10,80 nphi=2 phi=30,150 sza=60"],
         lambda r: r[-1:], keep_stdout=False)

    # --- bidirectional surfaces (LAMBER off; BDREF spectra.f:249-296, SURFAC's quadrature disort.f:3765-3912): the
    #     ocean model (wind 5 m/s, pigment 0.1 mg/m3), Hapke's soil, Ross-thick / Li-sparse; radiances up and down +
    #     fluxes, a thermal + solar wavelength over the ocean, a flux-only NSTR 16 run.  The records carry the model
    #     parameters and, for the ocean, the water's refractive index and sub-surface reflectance at the wavelength
    emit("brdf_surfaces",
         ["idatm=4 isat=0 wlinf=.5 wlsup=.9 wlinc=.2 isalb=7 sc=0.1,5,34.3,0 nstr=8 iout=20 nzen=5 uzen=0,80 nphi=3 phi=0,180 sza=40",
          "idatm=4 isat=0 wlinf=.5 wlsup=.9 wlinc=.2 isalb=8 sc=0.6,0.3,0.4,0.1 nstr=8 iout=21 nzen=5 uzen=100,180 nphi=3 phi=0,180 sza=40",
          "idatm=4 isat=0 wlinf=.5 wlsup=.9 wlinc=.2 isalb=9 sc=0.08,0.03,0.0005,1.0,2.0 nstr=8 iout=20 nzen=5 uzen=0,80 nphi=3 phi=0,180 sza=40",
          "idatm=2 isat=0 wlinf=3.7 wlsup=3.9 wlinc=.1 isalb=7 sc=1.0,10,34.3,0 nstr=12 iout=20 nzen=4 uzen=0,75 nphi=2 phi=20,160 sza=55 tcloud=1 zcloud=3",
          "idatm=4 isat=0 wlinf=.4 wlsup=1.0 wlinc=.3 isalb=8 sc=0.7,0.2,0.3,0.2 nstr=16 iout=10 sza=30",
          "idatm=4 isat=0 wlinf=.6 wlsup=.6 isalb=9 sc=0.08,0.03,0.0005,1.0,2.0 nstr=4 iout=10 sza=70"],
         lambda r: r[::2], keep_stdout=False)

    # --- intensities at the quadrature angles (USRANG = false: CMPINT, disort.f:1658-1778).  SBDART never asks for
    #     them, so the reference's DISORT is called directly (oracle/_ref/disort_ref_cli) on inputs of the records
    #     above with ONLYFL and USRANG switched off and three azimuths
    if not only or "quadangles_nstr16_4" in only:
        import dataclasses
        import numpy as np
        from sbdart_amd.records import F_ONLYFL, F_USRANG
        src = (read_records(os.path.join(HERE, "cfgB_sw_nstr16.sbdrec"))[::8]
               + read_records(os.path.join(HERE, "cfg3_lw_nstr16_cloud.sbdrec"))[::12]
               + read_records(os.path.join(HERE, "sbchk1.sbdrec"))[::60])
        recs = [dataclasses.replace(r.inputs_only(), flags=(r.flags & ~F_ONLYFL & ~F_USRANG), phi=np.array([0.0, 70.0, 180.0]),
                                    umu=np.zeros(0), phi0=20.0) for r in src]
        with tempfile.TemporaryDirectory() as d:
            write_records(os.path.join(d, "in.sbdrec"), recs, with_out=False)
            subprocess.run([os.path.join(ROOT, "oracle", "_ref", "disort_ref_cli"), "in.sbdrec", "out.sbdrec", "1"],
                           cwd=d, check=True, capture_output=True)
            out = read_records(os.path.join(d, "out.sbdrec"))
        path = os.path.join(HERE, "quadangles_nstr16_4.sbdrec")
        write_records(path, out, with_out=True)
        manifest["quadangles_nstr16_4"] = {"namelists": ["(disort_ref_cli on the inputs of cfgB / cfg3 / sbchk1 records, ONLYFL and USRANG off)"],
                                           "records": len(out), "bytes": os.path.getsize(path),
                                           "sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(), "reference_warnings": []}
        print("quadangles_nstr16_4", len(out), "records")

    # --- IBCND = 1 (ALBTRN, disort.f:6718-7432): albedo and transmissivity of the whole medium for beam incidence at
    #     user angles (USRANG on, ONLYFL off: with both on the reference overruns its UMU array) or at the quadrature
    #     angles, over surfaces of albedo 0 / 0.3 / 0.8, plus a single thick layer.  SBDART never sets IBCND, so the
    #     reference's DISORT is called directly (oracle/_ref/disort_ref_cli)
    if not only or "albtrn_ibcnd1" in only:
        import dataclasses
        import numpy as np
        from sbdart_amd.records import F_LAMBER, F_ONLYFL, F_USRANG
        src = (read_records(os.path.join(HERE, "cfgB_sw_nstr16.sbdrec"))[::9]
               + read_records(os.path.join(HERE, "sbchk1.sbdrec"))[::70]
               + read_records(os.path.join(HERE, "cfg3_lw_nstr16_cloud.sbdrec"))[::15])
        recs = []
        for k, r in enumerate(src):
            fl = (F_LAMBER | F_USRANG) if k % 3 else (F_LAMBER | F_ONLYFL)
            recs.append(dataclasses.replace(r.inputs_only(), flags=fl, ibcnd=1, albedo=[0.0, 0.3, 0.8][k % 3],
                                            umu=np.array([0.1, 0.5, 0.9, 1.0]) if fl & F_USRANG else np.zeros(0),
                                            phi=np.zeros(0)))
        r = recs[1]
        recs.append(dataclasses.replace(r, nlyr=1, dtauc=r.dtauc[-1:] * 50, ssalb=r.ssalb[-1:], pmom=r.pmom[-1:],
                                        temper=r.temper[-2:], albedo=0.4))
        with tempfile.TemporaryDirectory() as d:
            write_records(os.path.join(d, "in.sbdrec"), recs, with_out=False)
            subprocess.run([os.path.join(ROOT, "oracle", "_ref", "disort_ref_cli"), "in.sbdrec", "out.sbdrec", "1"],
                           cwd=d, check=True, capture_output=True)
            out = read_records(os.path.join(d, "out.sbdrec"))
        path = os.path.join(HERE, "albtrn_ibcnd1.sbdrec")
        write_records(path, out, with_out=True)
        manifest["albtrn_ibcnd1"] = {"namelists": ["(disort_ref_cli with IBCND = 1 on the inputs of cfgB / sbchk1 / cfg3 records)"],
                                     "records": len(out), "bytes": os.path.getsize(path),
                                     "sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(), "reference_warnings": []}
        print("albtrn_ibcnd1", len(out), "records")

    # --- ill-conditioned on purpose (kept apart from the 5e-6 parity files): the thermal window on a 65-level
    #     regridded atmosphere -- dozens of layers of optical depth ~1e-6 make the boundary-value system so
    #     nearly singular that the reference's own answer moves by 3e-5 when its arithmetic is merely contracted
    #     to fused multiply-adds (tests/test_gpu_parity.py::test_ill_conditioned_records)
    emit("illcond/thin65_thermal",
         ["idatm=4 wlinf=8 wlsup=9.6 wlinc=4.16667 iday=200 time=0 alat=35 alon=-120 nf=2 uo3=0.35 ngrid=65 "
          "zgrid1=2 zgrid2=10 iout=1 nstr=16"],
         lambda r: [x for x in r if x.kd == 3][::9][:6], keep_stdout=False)

    # --- layers a few ulps from conservative scattering (molecular scattering with a trace of absorption: SSALB = 1 - 1e-16
    #     .. 1 - 2e-15, which DISORT does not dither): two runs of the end-to-end fuzz of round 5 whose items came back NaN
    #     from the engine (tests/test_gpu_parity.py::test_rayleigh_layer_next_to_conservative).  The user's data files are
    #     the ones tests/test_band_model.py writes.
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_band_model import USER_FILES  # noqa: E402

    def near_conservative(recs, n_near, with_exact_one):
        near = [r for r in recs if 0 < 1 - r.ssalb.max() < 1e-13][:n_near]
        one = [r for r in recs if r.ssalb.max() == 1.0][:1] if with_exact_one else []
        plain = [r for r in recs if 1 - r.ssalb.max() > 1e-6][:1]
        return near + one + plain
    emit("illcond/rayleigh_next_to_conservative",
         ["idatm=3 csza=0.5 uw=2 sclh2o=2.5 uw=1.5 zpres=0.5 isalb=9 sc=0.2,0.01,0.002,1.0,2.0 ngrid=65 zgrid1=1 zgrid2=30 "
          "nothrm=1 iout=6 nstr=16 nzen=5 uzen=0,80 nphi=2 phi=0,180 zout=0,100 wlinc=.01 isat=-1"],
         lambda r: near_conservative(r, 3, True), keep_stdout=False, files=USER_FILES)
    emit("illcond/nstr40_next_to_conservative",
         ["idatm=5 wlinf=0.3 wlsup=0.45 wlinc=0 csza=0.2 uo3=0.2 xo4=0 xn2o=0.1 iout=7 nstr=40 nre=0"],
         lambda r: [x for x in r if x.iwl in (16, 17, 22)][:3] + [x for x in r if x.iwl == 1][:1], keep_stdout=False,
         files=USER_FILES)

    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("total bytes", sum(e["bytes"] for e in manifest.values()))


if __name__ == "__main__":
    main()
