"""Compact-form host entry point (sbd_fleet_solve_mix_host) on the bench's sweep for several first-pass sizes / growth
factors (SBD_HOST_FIRST_PASS, SBD_HOST_PASS_GROWTH): where the gap to the resident-input rate comes from.  GPU box."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
torch.cuda.init()
from sbdart_amd.engine import DisortFleet
from sbdart_amd.workload import sw_sweep_mix

mx = sw_sweep_mix(nwl=49152, nstr=16, nlyr=33, seed=12345)
pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()
m_in = [x if isinstance(x, tuple) else pin(x) for x in mx.mix_args()]
m_w = pin(mx.weight)
for first, grow in [(None, None), (8192, 1.3), (8192, 2.0), (4096, 2.0), (16384, 2.0), (21876, 1.0), (11000, 1.0), (5469, 4.0)]:
    for k, v in (("SBD_HOST_FIRST_PASS", first), ("SBD_HOST_PASS_GROWTH", grow)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    fleet = DisortFleet(nlyr=mx.nlyr, nstr=mx.nstr, nmom=mx.nmom, temper=mx.temper, umu0=mx.umu0, btemp=mx.btemp,
                        ttemp=mx.ttemp, temis=mx.temis, onlyfl=True, level_out=[0, mx.nlyr], devices=[0])
    step = lambda: fleet.solve_mix(*m_in, weight=m_w, items=False)[3]
    step()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    fleet.close()
    print(json.dumps({"first": first, "grow": grow, "ms_median": 1e3 * float(np.median(ts)), "ms_min": 1e3 * min(ts)}), flush=True)
