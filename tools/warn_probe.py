#!/usr/bin/env python3
"""One INPUT on the GPU box: which work items raise the engine's errmsg-2 stand-in (status bit 1) while the oracle
(bit-equal to the reference) does not raise the RCOND warning -- and what those items look like.
   python tools/warn_probe.py "idatm=2 wlinf=1 ..." """
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
print(len(recs), "records; nlyr", recs[0].nlyr, "nstr", recs[0].nstr)
flux, uu, st = solve_records(recs)
st = np.asarray(st)
for i, r in enumerate(recs):
    o = pyoracle.disort(r)
    if (int(st[i]) & 7) or (o["status"] & ~0):
        if (int(st[i]) & 7) == 0 and o["status"] == 0: continue
        print("rec %d wl %.4f kd %d engine st %d oracle st %d fbeam %.3g" % (i, r.wl, r.kd, int(st[i]), o["status"], r.fbeam))
        print("   dtauc", np.array2string(np.asarray(r.dtauc), precision=3, max_line_width=200))
        print("   ssalb", np.array2string(np.asarray(r.ssalb), precision=6, max_line_width=200))
        print("   flup", flux[i][2][:2], "oracle", o["flup"][:2])
