import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle
from sbdart_amd.engine import engine_for_record
from sbdart_amd.records import read_records
np.set_printoptions(linewidth=220, precision=5)
name = sys.argv[1] if len(sys.argv) > 1 else "sbchk2"
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
recs = read_records(os.path.join(ROOT, "tests", "golden", name + ".sbdrec"))
r = recs[idx]
o = pyoracle.disort(r, debug_mode=mode)
d = o["dbg"]
n, L, nn = r.nstr, r.nlyr, r.nstr // 2
with engine_for_record(r) as eng:
    flux, uu, st = eng.solve(r.dtauc[None], r.ssalb[None], r.pmom[None], [r.wvnmlo], [r.wvnmhi], [r.fbeam], [r.albedo], [r.plank])
    def fetch(which, per, shape, wid):
        buf = np.zeros(per * (mode + 1))
        got = eng._L.sbd_engine_debug_copy(eng._h, wid, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
        assert got == buf.nbytes, got
        return buf[per * mode:].reshape(shape)
    g = dict(gc=fetch("gc", L*n*n, (L, n, n), 0), kk=fetch("kk", L*n, (L, n), 1), zz=fetch("zz", L*n, (L, n), 3),
             zplk0=fetch("zp0", L*n, (L, n), 4), zplk1=fetch("zp1", L*n, (L, n), 5), ll=fetch("ll", L*n, (L, n), 6))
print("status", st, "flux err", [float(np.abs(flux[0][c] - getattr(r, f)).max()) for c, f in enumerate(("rfldir","rfldn","flup","dfdt","uavg"))])
for k in ("kk", "gc", "zz", "zplk0", "zplk1", "ll"):
    a, b = g[k], d[k]
    err = np.abs(a - b).reshape(L, -1).max(axis=1)
    sc = np.abs(b).reshape(L, -1).max(axis=1) + 1e-300
    print(f"{k:6s} worst abs {err.max():.3e} rel {np.max(err/sc):.3e} at layer {int(np.argmax(err/sc))+1}")
lc = int(sys.argv[4]) if len(sys.argv) > 4 else 1
print("kk gpu", g["kk"][lc-1]); print("kk ora", d["kk"][lc-1])
print("gc gpu\n", g["gc"][lc-1][:4, :8]); print("gc ora\n", d["gc"][lc-1][:4, :8])
print("zz gpu", g["zz"][lc-1]); print("zz ora", d["zz"][lc-1])
print("ll gpu", g["ll"][lc-1]); print("ll ora", d["ll"][lc-1])
