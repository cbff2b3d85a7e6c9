#!/usr/bin/env python3
"""Side benchmark (not the headline): radiance mode, BASELINE config 4 shape
(nstr=32, 20 zenith x 16 azimuth angles, 33 layers), synthetic optical properties."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sbdart_amd.engine import DisortEngine
from sbdart_amd.workload import sw_sweep
nwl = int(sys.argv[1]) if len(sys.argv) > 1 else 768
nstr = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sw = sw_sweep(nwl=nwl, nstr=nstr, thermal_above_um=99.0)
if len(sys.argv) > 3 and sys.argv[3] == "rayleigh":      # clear sky: molecular scattering alone (moments 1, 0, 0.1, 0, ...)
    sw.pmom[...] = 0.0
    sw.pmom[..., 0] = 1.0
    sw.pmom[..., 2] = 0.1
uzen = np.linspace(0, 85, 20)
umu = np.cos(np.deg2rad(uzen[::-1]))
phi = np.linspace(0, 180, 16)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
eng = DisortEngine(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp,
                   ttemp=sw.ttemp, temis=0.0, onlyfl=False, umu=umu, phi=phi, level_out=[0, sw.nlyr])
ins = [t(sw.dtauc), t(sw.ssalb), t(sw.pmom), t(sw.wvnmlo), t(sw.wvnmhi), t(sw.fbeam), t(sw.albedo), t(sw.plank)]
eng.solve(*ins); torch.cuda.synchronize()
eng.enable_timing(True)
t0 = time.perf_counter(); f, uu, st = eng.solve(*ins); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"radiance nstr={nstr}: {sw.nwork} solves ({nwl} spectral points) x {nstr} azimuth modes in {dt*1e3:.1f} ms -> "
      f"{nwl/dt:.0f} spectral-points/s, chunk {eng.chunk}, phases ms {[round(eng.last_ms(p),2) for p in range(5)]}, "
      f"status!=0: {int((st!=0).sum())}, finite {bool(torch.isfinite(uu).all())}")
