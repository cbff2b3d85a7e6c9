#!/usr/bin/env python3
"""A/B of TERPEV on the matrix cores (sbd_terpev.hpp) against the layer kernel's own: the cfgC workload's intensities with
both, in one process (the switch is read at engine creation).   python tools/terpev_ab.py [NWL]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
from sbdart_amd.engine import DisortEngine
from sbdart_amd.workload import sw_sweep
nwl = int(sys.argv[1]) if len(sys.argv) > 1 else 384
sw = sw_sweep(nwl=nwl, seed=12345, nstr=32, nlyr=33, thermal_above_um=99.0)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ins = [t(sw.dtauc), t(sw.ssalb), t(sw.pmom), t(sw.wvnmlo), t(sw.wvnmhi), t(sw.fbeam), t(sw.albedo), t(sw.plank)]
res = {}
for name, env in (("layer kernel", "1"), ("matrix cores", None)):
    if env: os.environ["SBD_NO_TERPEV_MFMA"] = env
    else: os.environ.pop("SBD_NO_TERPEV_MFMA", None)
    eng = DisortEngine(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp, ttemp=sw.ttemp,
                       temis=0.0, level_out=[0, sw.nlyr], device=0, onlyfl=False,
                       umu=np.cos(np.deg2rad(np.linspace(0, 85, 20)[::-1])), phi=np.linspace(0, 180, 16))
    eng.solve(*ins); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        flux, uu, st = eng.solve(*ins)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    res[name] = (flux.cpu().numpy(), uu.cpu().numpy(), st.cpu().numpy(), dt)
    print(f"{name}: {1e3 * dt:.3f} ms per step, {sw.nwl / dt:.0f} points/s, nonzero status {int((res[name][2] != 0).sum())}")
    eng.close()
a, b = res["layer kernel"], res["matrix cores"]
sc = np.abs(a[1]).max(axis=(1, 2, 3), keepdims=True) + 1e-300
print("intensities: worst |difference| / item maximum = %.3e ; fluxes identical: %s ; status equal: %s ; finite: %s"
      % (float((np.abs(a[1] - b[1]) / sc).max()), bool(np.array_equal(a[0], b[0])), bool(np.array_equal(a[2], b[2])), bool(np.isfinite(b[1]).all())))
