"""cfgD (NSTR 32 x 50 layers, 6 144 points = 16 258 solves) with the batch in one pass or cut into several
(SBD_CHUNK): do the layer and the band kernel of NSTR 32 gain from running beside each other on two streams?"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
torch.cuda.init()
from sbdart_amd.engine import DisortEngine
from sbdart_amd.workload import sw_sweep
sw = sw_sweep(nwl=int(sys.argv[1]) if len(sys.argv) > 1 else 6144, nstr=32, nlyr=50, seed=12345)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ins = [t(sw.dtauc), t(sw.ssalb), t(sw.pmom), t(sw.wvnmlo), t(sw.wvnmhi), t(sw.fbeam), t(sw.albedo), t(sw.plank)]
for chunk in (None, 8192, 5500, 4096, 2800):
    if chunk is None: os.environ.pop("SBD_CHUNK", None)
    else: os.environ["SBD_CHUNK"] = str(chunk)
    eng = DisortEngine(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp, ttemp=sw.ttemp,
                       temis=sw.temis, onlyfl=True, level_out=[0, sw.nlyr], device=0)
    eng.solve(*ins); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); eng.solve(*ins); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(json.dumps({"chunk": chunk, "engine_chunk": eng.chunk, "ms_median": 1e3 * float(np.median(ts)), "points_per_s": sw.nwl / float(np.median(ts))}), flush=True)
    eng.close()
