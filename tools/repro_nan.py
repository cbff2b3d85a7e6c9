"""GPU box: the work items of one INPUT (Fortran host, SBD_DUMP_OPTICS) through the engine, item by item against the oracle:
which item goes wrong, in which array, with which status."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import torch; torch.cuda.init()
import pyoracle
from sbdart_amd.engine import solve_records
from sbdart_amd.records import read_records
from test_band_model import USER_FILES
nl = sys.argv[1]
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "INPUT"), "w").write("\n &INPUT\n" + nl + "\n /\n")
    for name, text in USER_FILES.items():
        open(os.path.join(d, name), "w").write(text)
    out = os.path.join(d, "items.sbdrec")
    subprocess.run([os.path.join(ROOT, "sbdart_amd", "bin", "sbdart_amd")], cwd=d, env=dict(os.environ, SBD_DUMP_OPTICS=out, SBD_OPTICS=os.path.join(d, "none")), capture_output=True)
    recs = [r for r in read_records(out) if r.ff != 0.0]
for r in recs:                       # (the dump of the host's band model does not carry the LAMBER bit)
    if getattr(r, "ibdrf", 0) == 0:
        r.flags |= 4
print(len(recs), "items; nstr", recs[0].nstr, "nlyr", recs[0].nlyr, "onlyfl", recs[0].onlyfl, "ibdrf", getattr(recs[0], "ibdrf", 0))
for lev in (None, [0, recs[0].nlyr]):
    flux, uu, st = solve_records(recs, level_out=lev)
    for i, r in enumerate(recs):
        bad = [f for c, f in enumerate(("rfldir", "rfldn", "flup", "dfdt", "uavg")) if not np.isfinite(flux[i][c]).all()]
        ubad = uu[i] is not None and not np.isfinite(uu[i]).all()
        if bad or ubad or st[i]:
            o = pyoracle.disort(r)
            print("levels", lev, "item", i, "wl", r.wl, "status", st[i], "oracle status", o["status"], "non-finite:", bad, "uu" if ubad else "",
                  "oracle finite:", all(np.isfinite(o[f]).all() for f in ("rfldn", "flup")), "fbeam", r.fbeam, "ssalb max", r.ssalb.max(), "tau", r.dtauc.sum())
            if ubad:
                w = np.argwhere(~np.isfinite(uu[i]))
                print("   uu non-finite at (phi, level, mu):", w[:6].tolist(), "of", uu[i].shape)

# which workspace array of the bad item goes non-finite first (all levels: stored-factor path keeps everything)
from sbdart_amd.engine import engine_for_record
bad_items = [i for i, r in enumerate(recs) if not np.isfinite(solve_records([r])[0][0]).all()]
for i in bad_items[:2]:
    r = recs[i]
    o = pyoracle.disort(r, debug_mode=0)
    with engine_for_record(r, level_out=None) as eng:
        eng.solve(r.dtauc[None], r.ssalb[None], r.pmom[None], [r.wvnmlo], [r.wvnmhi], [r.fbeam], [r.albedo], [r.plank])
        n, L = r.nstr, r.nlyr
        nmode = n                                               # radiance: every azimuth mode
        for which, name, per in ((1, "kk", n), (3, "zz", n), (0, "gc", n * n), (6, "ll", n)):
            a = eng.debug_array(which, np.float64, nmode * L * per).reshape(nmode, L, per)
            w = np.argwhere(~np.isfinite(a))
            print("item", i, name, "non-finite entries:", len(w), "first (mode, layer, k):", w[:4].tolist())
    sidx = np.argsort(r.ssalb)[-3:]
    print("   ssalb top:", [(int(k), float(1 - r.ssalb[k])) for k in sidx], "oracle kk min per layer:", float(np.abs(o["dbg"]["kk"]).min()))
    print("   layer", int(sidx[-1]), "dtauc", float(r.dtauc[sidx[-1]]), "pmom[:4]", r.pmom[sidx[-1]][:4].tolist(), "f", float(r.pmom[sidx[-1]][n]))
