"""GPU box: are the particular solutions of the layers the reference-algorithm layer kernel serves the ORACLE's, bit for
bit?  (UPBEAM next to a singular system, UPISOT in a nearly conservative thermal layer: sbd_layer.hpp forms GL, CC and
the systems without contraction and factors them by SGEFA's rule since round 5.)  Prints per case the largest
difference of ZZ / ZPLK0 / ZPLK1 in units of the column maximum and whether the arrays are identical."""
import dataclasses, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.init()
import pyoracle
from sbdart_amd.engine import engine_for_record
from sbdart_amd.records import F_LAMBER, F_ONLYFL, SolveRecord
from test_gpu_parity import _thermal_record

def probe(name, r):
    o = pyoracle.disort(r, debug_mode=0)
    d = o["dbg"]
    with engine_for_record(r, level_out=None) as eng:
        flux, _, st = eng.solve(r.dtauc[None], r.ssalb[None], r.pmom[None], [r.wvnmlo], [r.wvnmhi], [r.fbeam], [r.albedo], [r.plank])
        n, L = r.nstr, r.nlyr
        zz = eng.debug_array(3, np.float64, L * n).reshape(L, n)
        z0 = eng.debug_array(4, np.float64, L * n).reshape(L, n)
        z1 = eng.debug_array(5, np.float64, L * n).reshape(L, n)
        kk = eng.debug_array(1, np.float64, L * n).reshape(L, n)
    out = [name, "status gpu/oracle", int(st[0]), int(o["status"])]
    for nm, a, b in (("zz", zz, d["zz"]), ("zplk0", z0, d["zplk0"]), ("zplk1", z1, d["zplk1"]), ("kk", kk, d["kk"])):
        sc = np.abs(b).max(axis=1, keepdims=True) + 1e-300
        out += [nm, "identical" if np.array_equal(a, b) else f"{float((np.abs(a - b) / sc).max()):.2e}"]
    fl = max(float(np.abs(flux[0][c] - o[f]).max() / (np.abs(o[f]).max() + 1e-300)) for c, f in enumerate(("rfldir", "rfldn", "flup", "dfdt", "uavg")))
    print(*out, "flux", f"{fl:.2e}", flush=True)

for nstr in (8, 16):
    nmom = nstr + 2
    g = np.array([0.7, 0.8, 0.6])
    base = SolveRecord(nlyr=3, nstr=nstr, nmom=nmom, flags=F_LAMBER | F_ONLYFL, wvnmlo=10000.0, wvnmhi=10100.0, fbeam=1.0, umu0=0.5,
                       phi0=0.0, albedo=0.2, btemp=290.0, ttemp=0.0, temis=0.0, dtauc=np.array([0.2, 0.7, 0.4]),
                       ssalb=np.array([0.6, 0.9, 0.8]), temper=np.linspace(220.0, 290.0, 4),
                       pmom=g[:, None] ** np.arange(nmom + 1)[None, :], umu=np.zeros(0), phi=np.zeros(0))
    kk = pyoracle.disort(base, debug_mode=0)["dbg"]["kk"]
    for lc in range(3):
        for k in kk[lc][nstr // 2:]:
            if 1.02 < k < 20.0:
                for delta in (1e-11, 1e-6):
                    probe(f"beam nstr {nstr} layer {lc} k {k:.4f} delta {delta:g}", dataclasses.replace(base, umu0=float(1.0 / (k * (1.0 + delta)))))
for nstr in (4, 8, 16, 32):
    for off in (1, 3, 22, 200, 5000):
        probe(f"thermal nstr {nstr} ssalb 1 - {off} ulps", _thermal_record(nstr, 3, 1.0, 1.0 - off * 2.0 ** -53))
    probe(f"thermal nstr {nstr} ssalb 1 (dithered)", _thermal_record(nstr, 3, 1.0, 1.0))
