#!/usr/bin/env python3
"""Developer timing of one shape under engine switches (bench.py refuses to run with them):
   python tools/bench_switch.py NSTR NLYR NWL  VAR=VALUE ...   -> ms per step and the phase times."""
import os, sys, time
nstr, nlyr, nwl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    os.environ[k] = v
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
from sbdart_amd.engine import DisortEngine
from sbdart_amd.workload import sw_sweep
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
for _ in range(2):
    eng.solve_device(*d_in, out=(flux, None, status), stream=s.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    eng.solve_device(*d_in, out=(flux, None, status), stream=s.cuda_stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0)/n
eng.enable_timing(True)
eng.solve_device(*d_in, out=(flux, None, status), stream=s.cuda_stream)
ph = [round(eng.last_ms(p), 2) for p in range(5)]
print(sys.argv[4:], "%.2f ms/step" % (dt*1e3), "%.0f pts/s" % (nwl/dt), "phases", ph, "bad", int((status != 0).sum()),
      "finite", bool(torch.isfinite(flux).all()), "fallback layers", eng.last_fallback_layers(), "of", W*nlyr)
