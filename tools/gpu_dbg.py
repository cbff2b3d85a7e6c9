import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sbdart_amd.engine import solve_records
from sbdart_amd.records import read_records
np.set_printoptions(linewidth=200, precision=6)
name = sys.argv[1] if len(sys.argv) > 1 else "sbchk2"
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
recs = read_records(os.path.join(ROOT, "tests", "golden", name + ".sbdrec"))
r = recs[idx]
flux, uu, st = solve_records([r])
print("status", st, "nstr", r.nstr, "plank", r.plank, "fbeam", r.fbeam, "umu0", r.umu0, "albedo", r.albedo)
for c, f in enumerate(("rfldir", "rfldn", "flup", "dfdt", "uavg")):
    ref = getattr(r, f)
    print(f, "gpu", flux[0][c][[0, 1, 2, -2, -1]], "ref", ref[[0, 1, 2, -2, -1]], "maxabs", np.abs(flux[0][c] - ref).max())
