import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
t0=time.perf_counter()
from sbdart_amd.engine import DisortFleet
from sbdart_amd.workload import sw_sweep
from sbdart_amd import _lib
sw = sw_sweep(nwl=751, nstr=16, seed=1)
L=_lib.load()
t1=time.perf_counter()
import ctypes as C
n=C.c_int(0)
hip=C.CDLL("libamdhip64.so")
ta=time.perf_counter(); hip.hipInit(0); tb=time.perf_counter(); hip.hipGetDeviceCount(C.byref(n)); tc=time.perf_counter()
print("hipInit %.3f hipGetDeviceCount %.3f"%(tb-ta, tc-tb))
kw=dict(nlyr=sw.nlyr,nstr=sw.nstr,nmom=sw.nmom,temper=sw.temper,umu0=sw.umu0,btemp=sw.btemp,ttemp=sw.ttemp,temis=sw.temis,onlyfl=True,level_out=[0,sw.nlyr],devices=[0],max_batch=sw.nwork)
for rep in range(3):
    t2=time.perf_counter(); fl=DisortFleet(**kw); t3=time.perf_counter()
    ins=(sw.dtauc,sw.ssalb,sw.pmom,sw.wvnmlo,sw.wvnmhi,sw.fbeam,sw.albedo,sw.plank)
    fl.solve(*ins,weight=sw.weight,items=False); t4=time.perf_counter()
    fl.solve(*ins,weight=sw.weight,items=False); t5=time.perf_counter()
    fl.close(); t6=time.perf_counter()
    print("rep %d: create %.4f first solve %.4f second solve %.4f close %.4f"%(rep,t3-t2,t4-t3,t5-t4,t6-t5))
