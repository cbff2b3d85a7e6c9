#!/usr/bin/env python3
"""Developer tool: where the RADIANCE variant of the fast layer kernel spends its shader-clock ticks.
Needs sbd_k_layer2r.hip built with -DSBD_PHASE_TICKS (see tools/layer_phases.py):
   python tools/layer_phases_rad.py NSTR NWL"""
import ctypes, os, sys
nstr, nwl = int(sys.argv[1]), int(sys.argv[2])
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
from sbdart_amd import _lib
from sbdart_amd.engine import DisortEngine
from sbdart_amd.workload import sw_sweep
L = _lib.load()
sw = sw_sweep(nwl=nwl, nstr=nstr, thermal_above_um=99.0)
umu = np.cos(np.deg2rad(np.linspace(0, 85, 20)[::-1]))
phi = np.linspace(0, 180, 16)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
eng = DisortEngine(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp,
                   ttemp=sw.ttemp, temis=0.0, onlyfl=False, umu=umu, phi=phi, level_out=[0, sw.nlyr])
ins = [t(sw.dtauc), t(sw.ssalb), t(sw.pmom), t(sw.wvnmlo), t(sw.wvnmhi), t(sw.fbeam), t(sw.albedo), t(sw.plank)]
buf = (ctypes.c_ulonglong*16)()
eng.solve(*ins); torch.cuda.synchronize()
assert L.sbd_debug_layer2r_ticks(buf, 1) == 0
eng.solve(*ins); torch.cuda.synchronize()
assert L.sbd_debug_layer2r_ticks(buf, 0) == 0
v = list(buf)
waves = max(v[7], 1)
names = ["GL + S+- + Q+-", "Cholesky x2", "B = C^T L", "Jacobi", "eigenvectors + outputs", "TERPEV + UPISOT", "UPBEAM + rest"]
tot = sum(v[:7])
for n, x in zip(names, v[:7]):
    print("%-24s %9.0f ticks/wave  %5.1f %%" % (n, x/waves, 100.0*x/tot))
print("%-24s %9.0f ticks/wave  %5.1f %%" % ("  of which TERPEV", v[9]/waves, 100.0*v[9]/tot))
print("waves %d, sweeps/wave %.2f, ticks/wave %.0f" % (waves, v[8]/waves, tot/waves))
