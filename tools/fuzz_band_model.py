#!/usr/bin/env python3
"""Random-INPUT fuzz of the host band model against the reference run live (oracle/_ref/sbdart_capture):
python tools/fuzz_band_model.py SEED COUNT -- every switch of the band model drawn at random, the work items
compared with the reference DISORT arguments (bar 1e-12, see tests/test_band_model.py).  Build-container tool."""
import sys, os, random, subprocess, tempfile, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/tests')
from test_band_model import host_items, reference_items, compare, aerosol_file
random.seed(int(sys.argv[1]) if len(sys.argv)>1 else 1)
def pick(*a): return random.choice(a)
bad=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 60):
    p=[]
    p.append("idatm=%d"%pick(1,2,3,4,5,6))
    lo=pick(.25,.3,.4,.55,1.,2.,3.5,5.,8.)
    hi=lo*pick(1.0,1.2,1.5,2.,4.)
    hi=min(hi,90.)
    p.append("wlinf=%g wlsup=%g"%(lo,hi))
    if hi>lo: p.append("wlinc=%g"%pick(0,.01*lo,-.01,-.003,(1e4/lo-1e4/hi)/pick(7,23,50) if (1e4/lo-1e4/hi)/50>1 else .02*lo))
    p.append(pick("sza=%g"%pick(0,20,45,60,75,85,89.995,95), "csza=%g"%pick(.2,.5,.9), "iday=%d time=%g alat=%g alon=%g"%(pick(10,100,200,355),pick(0,6,12,18,22.5),pick(-60,0,35,70),pick(-120,0,75))))
    if random.random()<.5: p.append("kdist=%d"%pick(0,1,2,3))
    if random.random()<.4: p.append("nf=%d"%pick(0,1,2,3))
    if random.random()<.3: p.append("uw=%g"%pick(.5,2,4))
    if random.random()<.3: p.append("uo3=%g"%pick(.2,.35))
    if random.random()<.2: p.append("sclh2o=%g uw=1.5"%pick(1.,2.5))
    if random.random()<.2: p.append("pbar=%g"%pick(900,1030))
    elif random.random()<.2: p.append("zpres=%g"%pick(.5,2.2))
    if random.random()<.3: p.append("xco2=%g xch4=%g"%(pick(280,420,800),pick(.8,1.8,3)))
    if random.random()<.2: p.append("xo4=%g xn2o=%g"%(pick(0,2),pick(.1,.4)))
    if random.random()<.45:
        c=pick("tcloud=%g zcloud=%g nre=%g"%(pick(.5,5,40),pick(.5,2,6,11),pick(4,8,20,-25,-60)),
               "lwp=%g zcloud=%g nre=%g"%(pick(20,150),pick(1,3),pick(6,12)),
               "tcloud=%g,%g zcloud=%g,-%g nre=%g,%g"%(pick(3,12),pick(1,3,.5),pick(1,2),pick(4,7),pick(6,10),pick(8,16)))
        p.append(c)
        if random.random()<.3: p.append("rhcld=%g krhclr=%d"%(pick(.8,1.),pick(0,1)))
        if random.random()<.3: p.append("imomc=%d"%pick(3,4,5))
    aerfile=None
    if random.random()<.12:                      # aerosol.dat: a full column (33 layers) at 1-5 wavelengths
        p.append("iaer=-1 imoma=%d"%pick(1,3,4))
        ws=sorted(random.sample([.2,.35,.5,.7,1.,1.6,2.5,4.,9.,20.],pick(1,2,3,5)))
        aerfile=aerosol_file(tuple(ws),33,pick(1,1,3,12),seed=it)
    elif random.random()<.4:
        p.append(pick("iaer=%d vis=%g"%(pick(1,2,3,4),pick(5,23,60)), "iaer=%d tbaer=%g rhaer=%g"%(pick(1,2,3,4),pick(.05,.5),pick(.3,.75,.9,.99))))
        if random.random()<.3: p.append("nosct=%d"%pick(1,3))
    if random.random()<.2: p.append("jaer=%d zaer=%g taerst=%g"%(pick(1,2,3,4),pick(15,22),pick(.01,.1)))
    if random.random()<.5: p.append(pick("albcon=%g"%pick(0,.3,.9),"isalb=%d"%pick(1,2,3,4,5,6),"isalb=10 sc=.25,.25,.25,.25",
                                         "isalb=%d sc=%g,%g,34.3,0"%(pick(7,-7),pick(0,.1,1.),pick(2,7,12)),
                                         "isalb=%d sc=%g,%g,%g,%g"%(pick(8,-8),pick(.4,.6,.8),pick(.1,.3),pick(0,.4),pick(.05,.1)),
                                         "isalb=%d sc=%g,%g,%g,1.0,2.0"%(pick(9,-9),pick(.05,.08,.2),pick(.01,.03),pick(.0005,.002))))
    if aerfile is None and random.random()<.25: p.append("ngrid=%d zgrid1=%g zgrid2=%g"%(pick(20,40,65),pick(.5,1,2),pick(10,30)))
    if random.random()<.2: p.append("nothrm=%d"%pick(0,1))
    if random.random()<.2: p.append("xrsc=%g"%pick(0,.5,2))
    rad=random.random()<.3
    if rad:
        p.append("iout=%d nstr=%d nzen=%d uzen=%s nphi=%d phi=0,%d"%(pick(5,6,20,21,22,23),pick(4,8,16),pick(2,5),pick("0,80","100,175","10,170"),pick(2,3),pick(90,180)))
        if random.random()<.4: p.append("corint=t")
    else:
        p.append("iout=%d nstr=%d"%(pick(1,7,10,11),pick(4,8,16)))
    if random.random()<.15: p.append("isat=%d"%pick(1,4,9,13,17,22,26))
    nl=" ".join(p)
    with tempfile.TemporaryDirectory() as d:
        if aerfile:
            for sub in ("/r","/m"):
                os.makedirs(d+sub); open(d+sub+"/aerosol.dat","w").write(aerfile)
        try:
            ref=reference_items(d+"/r",nl)
        except Exception as e:
            continue
        if not ref: continue
        try:
            mine=host_items(d+"/m",nl)
            w=compare(mine,ref,True)
            print("ok %5d items %.1e :: %s"%(len(ref),w,nl))
        except AssertionError as e:
            bad+=1
            print("FAIL ::",nl,"::",str(e)[:300])
print("failures",bad)
