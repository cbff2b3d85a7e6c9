"""Host entry point (sbd_fleet_solve_host, pinned inputs, moments per spectral point) on the bench's sweep for several
pass sizes (SBD_CHUNK): where the gap to the resident-input rate comes from.  Run on the GPU box."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
torch.cuda.init()
from sbdart_amd.engine import DisortFleet
from sbdart_amd.workload import sw_sweep

sw = sw_sweep(nwl=49152, nstr=16, nlyr=33, seed=12345, shard=0)
pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()
rows = np.ascontiguousarray(sw.wl_of, dtype=np.int32)
first = np.concatenate([[0], np.nonzero(np.diff(rows))[0] + 1])
h = dict(dt=pin(sw.dtauc), ss=pin(sw.ssalb), pm=pin(sw.pmom[first]), lo=pin(sw.wvnmlo), hi=pin(sw.wvnmhi),
         fb=pin(sw.fbeam), al=pin(sw.albedo), pl=pin(sw.plank), w=pin(sw.weight), rows=pin(rows))
out = []
for chunk in [int(x) for x in (sys.argv[1:] or ["32768", "16384", "11000", "8192", "5500"])]:
    os.environ["SBD_CHUNK"] = str(chunk)
    fleet = DisortFleet(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp,
                        ttemp=sw.ttemp, temis=sw.temis, onlyfl=True, level_out=[0, sw.nlyr], devices=[0])
    step = lambda: fleet.solve(h["dt"], h["ss"], h["pm"], h["lo"], h["hi"], h["fb"], h["al"], h["pl"], weight=h["w"],
                               items=False, pmom_row=h["rows"])[3]
    step()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    fleet.close()
    r = {"chunk": chunk, "passes": -(-sw.nwork // chunk), "ms_median": 1e3 * float(np.median(ts)), "ms_min": 1e3 * min(ts)}
    print(json.dumps(r), flush=True)
    out.append(r)
