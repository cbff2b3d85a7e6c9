"""GPU box: which LAYER of a work item goes wrong -- the items of one INPUT (Fortran host, SBD_DUMP_OPTICS, the user data
files of tests/test_band_model.py in the run directory), item ITEM (default 15) through the engine with every level kept:
per layer the non-finite entries of GC and the smallest |k| against the oracle's.  The item is left in
gpurun_out/nan_item.sbdrec.   usage: python tools/nan_layers.py "namelist text" [ITEM]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import torch; torch.cuda.init()
import pyoracle
from sbdart_amd.engine import solve_records, engine_for_record
from sbdart_amd.records import read_records, write_records
from test_band_model import USER_FILES
nl = sys.argv[1]
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "INPUT"), "w").write("\n &INPUT\n" + nl + "\n /\n")
    for name, text in USER_FILES.items():
        open(os.path.join(d, name), "w").write(text)
    out = os.path.join(d, "items.sbdrec")
    subprocess.run([os.path.join(ROOT, "sbdart_amd", "bin", "sbdart_amd")], cwd=d, env=dict(os.environ, SBD_DUMP_OPTICS=out, SBD_OPTICS=os.path.join(d, "none")), capture_output=True)
    recs = [r for r in read_records(out) if r.ff != 0.0]
for r in recs:
    if getattr(r, "ibdrf", 0) == 0: r.flags |= 4
r = recs[int(sys.argv[2]) if len(sys.argv) > 2 else 15]
os.makedirs(ROOT + "/gpurun_out", exist_ok=True)
write_records(ROOT + "/gpurun_out/nan_item.sbdrec", [r])
o = pyoracle.disort(r, debug_mode=0)
n, L = r.nstr, r.nlyr
with engine_for_record(r, level_out=None) as eng:
    eng.solve(r.dtauc[None], r.ssalb[None], r.pmom[None], [r.wvnmlo], [r.wvnmhi], [r.fbeam], [r.albedo], [r.plank])
    print("fallback layers", eng.last_fallback_layers())
    kk = eng.debug_array(1, np.float64, L * n).reshape(L, n)
    gc = eng.debug_array(0, np.float64, L * n * n).reshape(L, n, n)
    for lc in range(L):
        bad = np.argwhere(~np.isfinite(gc[lc]))
        okk = np.sort(np.abs(o["dbg"]["kk"].reshape(L, n)[lc]))
        ekk = np.sort(np.abs(kk[lc]))
        if len(bad) or lc in (19, 20, 21):
            print("layer", lc, "ssalb", r.ssalb[lc], "1-ssalb", 1 - r.ssalb[lc], "dtauc", r.dtauc[lc], "pmom[:5]", r.pmom[lc][:5], "pmom[n]", r.pmom[lc][n] if r.pmom.shape[1] > n else None,
                  "bad gc", len(bad), "cols", sorted(set(bad[:, 1].tolist()))[:8], "| oracle |k| smallest", okk[:3], "engine", ekk[:3], "max rel kk diff", np.max(np.abs(okk - ekk) / okk))
