import os, sys, time, subprocess
code = r'''
import os, sys, time, numpy as np, torch, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
torch.cuda.init()
from sbdart_amd.engine import DisortEngine
from sbdart_amd.workload import sw_sweep
sw = sw_sweep(nwl=49152, nstr=16, nlyr=33, seed=12345, shard=0)
dev = torch.device("cuda", 0)
eng = DisortEngine(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp, ttemp=sw.ttemp, temis=sw.temis, onlyfl=True, level_out=[0, sw.nlyr], device=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d_in = [t(sw.dtauc), t(sw.ssalb), t(sw.pmom), t(sw.wvnmlo), t(sw.wvnmhi), t(sw.fbeam), t(sw.albedo), t(sw.plank)]
W = sw.nwork
flux = torch.empty((W, 5, eng.nlev), dtype=torch.float64, device=dev); status = torch.empty(W, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(dev); torch.cuda.set_stream(s)
for _ in range(3): eng.solve_device(*d_in, out=(flux, None, status), stream=s.cuda_stream)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): eng.solve_device(*d_in, out=(flux, None, status), stream=s.cuda_stream)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(os.environ.get("SBD_CHUNK"), eng.chunk, "%.2f ms" % (dt * 1e3), "%.3f M pts/s" % (49152 / dt / 1e6))
'''
for ch in (sys.argv[1:] or ("8192", "16384", "21876", "32814", "43752", "65627")):
    env = dict(os.environ, SBD_CHUNK=ch)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
