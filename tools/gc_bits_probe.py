"""GPU box: are GC and KK of the reference-algorithm layer kernel (sbd_layer.hpp: ASYMTX restated, no contraction) the
ORACLE's bit for bit?  That decides whether the band system's LINPACK condition estimate (errmsg 2) can be reproduced
exactly on the device (round 6: band_rcond_kernel).  Run with SBD_FORCE_EIG_FALLBACK=1 (every layer through that kernel).
Prints per record set: layers compared, layers whose KK / GC differ in any bit, the largest relative difference."""
import glob, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
torch.cuda.init()
import pyoracle
from sbdart_amd.engine import engine_for_record
from sbdart_amd.records import read_records

assert os.environ.get("SBD_FORCE_EIG_FALLBACK") == "1"
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "tests/golden/illcond/*.sbdrec"))) + [os.path.join(ROOT, "tests/golden", f) for f in ("sbchk1.sbdrec", "cfgB_sw_nstr16.sbdrec", "cfgD_nstr32_50ly.sbdrec", "conservative_thermal.sbdrec")]
for f in files:
    recs = [r for r in read_records(f) if r.lamber and not r.ibcnd][:12]
    nl = nk = ng = 0
    worst_k = worst_g = 0.0
    for r in recs:
        o = pyoracle.disort(r, debug_mode=0)
        if o["status"] & 0x38:
            continue
        n, L = r.nstr, r.nlyr
        with engine_for_record(r, level_out=None) as eng:
            _, _, st = eng.solve(r.dtauc[None], r.ssalb[None], r.pmom[None], [r.wvnmlo], [r.wvnmhi], [r.fbeam], [r.albedo], [r.plank])
            svi = eng.debug_array(8, np.int32, 8)
            ncut = int(svi[0])
            kk = eng.debug_array(1, np.float64, L * n).reshape(L, n)
            gc = eng.debug_array(0, np.float64, L * n * n).reshape(L, n, n)
        okk = o["dbg"]["kk"].reshape(L, n)
        ogc = o["dbg"]["gc"].reshape(L, n, n)
        for lc in range(ncut):
            nl += 1
            if not np.array_equal(kk[lc], okk[lc]):
                nk += 1
                worst_k = max(worst_k, float(np.max(np.abs(kk[lc] - okk[lc]) / (np.abs(okk[lc]) + 1e-300))))
            a = gc[lc]
            if not (np.array_equal(a, ogc[lc]) or np.array_equal(a.T, ogc[lc])):
                ng += 1
                d1 = np.max(np.abs(a - ogc[lc])) / np.max(np.abs(ogc[lc]))
                d2 = np.max(np.abs(a.T - ogc[lc])) / np.max(np.abs(ogc[lc]))
                worst_g = max(worst_g, float(min(d1, d2)))
    print(f"{os.path.basename(f)}: {len(recs)} records, {nl} layers; KK differs in {nk} (max rel {worst_k:.2e}); GC differs in {ng} (max rel to the layer's max {worst_g:.2e})", flush=True)
