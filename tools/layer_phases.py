#!/usr/bin/env python3
"""Developer tool: where the fast layer kernel spends its shader-clock ticks, phase by phase.
Needs the layer kernel built with -DSBD_PHASE_TICKS:
   rm sbdart_amd/csrc/build/sbd_k_layer2f.o; make -C sbdart_amd/csrc HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -DSBD_PHASE_TICKS"
(and rebuilt without it afterwards: the tick atomics cost time)
   python tools/layer_phases.py NSTR NLYR NWL"""
import ctypes, os, sys
nstr, nlyr, nwl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
from sbdart_amd import _lib
from sbdart_amd.engine import DisortEngine
from sbdart_amd.workload import sw_sweep
L = _lib.load()
sw = sw_sweep(nwl=nwl, nstr=nstr, nlyr=nlyr, seed=12345, shard=0)
dev = torch.device("cuda", 0)
eng = DisortEngine(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp,
                   ttemp=sw.ttemp, temis=sw.temis, onlyfl=True, level_out=[0, sw.nlyr], device=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d_in = [t(sw.dtauc), t(sw.ssalb), t(sw.pmom), t(sw.wvnmlo), t(sw.wvnmhi), t(sw.fbeam), t(sw.albedo), t(sw.plank)]
W = sw.nwork
flux = torch.empty((W, 5, eng.nlev), dtype=torch.float64, device=dev)
status = torch.empty(W, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(dev)
torch.cuda.set_stream(s)
buf = (ctypes.c_ulonglong*16)()
eng.solve_device(*d_in, out=(flux, None, status), stream=s.cuda_stream)
torch.cuda.synchronize()
assert L.sbd_debug_layer2_ticks(buf, 1) == 0
eng.solve_device(*d_in, out=(flux, None, status), stream=s.cuda_stream)
torch.cuda.synchronize()
assert L.sbd_debug_layer2_ticks(buf, 0) == 0
v = list(buf)
waves = max(v[7], 1)
names = ["GL + S+- + Q+-", "Cholesky x2", "B = C^T L", "Jacobi", "eigenvectors + outputs", "UPISOT", "UPBEAM + rest"]
tot = sum(v[:7])
for n, x in zip(names, v[:7]):
    print("%-24s %9.0f ticks/wave  %5.1f %%" % (n, x/waves, 100.0*x/tot))
print("waves %d, sweeps/wave %.2f, ticks/wave %.0f" % (waves, v[8]/waves, tot/waves))
