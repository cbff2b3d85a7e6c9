#!/usr/bin/env python3
"""One INPUT on the GPU box, taken apart: the reference's DISORT records against the engine (per array, next to the
reference's own FMA sensitivity), then the two stdouts line by line where their numbers differ.
   python tools/dbg_case.py "idatm=... iout=11"        """
import sys, os, tempfile, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests'); sys.path.insert(0, ROOT + '/oracle')
from test_fortran_host import run_reference_and_host
from sbdart_amd.records import read_records
from sbdart_amd.engine import solve_records
import pyoracle
nl = sys.argv[1]
with tempfile.TemporaryDirectory() as d:
    ref, got, cap = run_reference_and_host(nl, d, from_input=True)
    recs = read_records(cap)
print(len(recs), "records; nlyr", recs[0].nlyr, "nstr", recs[0].nstr, "ibdrf", recs[0].ibdrf)
flux, uu, st = solve_records(recs)
names = ('rfldir', 'rfldn', 'flup', 'dfdt', 'uavg')
worst = []
for i, r in enumerate(recs):
    tw = pyoracle.disort(r, perturbed=True)
    for c, f in enumerate(names):
        refv = getattr(r, f); sc = np.abs(refv).max()
        if sc == 0: continue
        worst.append((np.abs(flux[i][c] - refv).max() / sc, np.abs(tw[f] - refv).max() / sc, i, f, r.wl, r.kd, int(st[i])))
worst.sort(reverse=True)
for w in worst[:6]: print("err %.2e  ref-sens %.2e  rec %d %s wl %.4f kd %d st %d" % w)
rl, gl = ref.splitlines(), got.splitlines()
print(len(rl), len(gl), "lines")
shown = 0
for a, b in zip(gl, rl):
    if a.split() != b.split():
        print("ref:", b); print("got:", a); shown += 1
        if shown > 12: break
