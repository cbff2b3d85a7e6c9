#!/usr/bin/env python3
"""Developer probe (GPU box): HIP engine vs golden reference records + C oracle, per file."""
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from sbdart_amd.engine import solve_records  # noqa: E402
from sbdart_amd.records import read_records  # noqa: E402

FLUX = ("rfldir", "rfldn", "flup", "dfdt", "uavg")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.sbdrec")))
bad = 0
for p in files:
    recs = read_records(p)
    t = time.time()
    try:
        flux, uu, st = solve_records(recs)
    except Exception as ex:  # noqa: BLE001
        print(os.path.basename(p), "EXCEPTION", ex)
        bad += 1
        continue
    dt = time.time() - t
    worst, wuu, wi = 0.0, 0.0, None
    nan = 0
    for i, r in enumerate(recs):
        for c, f in enumerate(FLUX):
            ref = getattr(r, f)
            sc = max(np.abs(ref).max(), 1e-300)
            d = np.abs(flux[i][c] - ref).max() / sc
            if not np.isfinite(d):
                nan += 1
                d = np.inf
            if d > worst:
                worst, wi = d, (i, f)
        if not r.onlyfl:
            d = np.abs(uu[i] - r.uu).max() / np.abs(r.uu).max()
            if not np.isfinite(d):
                nan += 1
                d = np.inf
            wuu = max(wuu, d)
    stbits = 0
    for s in st:
        stbits |= s
    print(f"{os.path.basename(p):28s} n={len(recs):4d} nstr={recs[0].nstr:2d} worst flux rel-to-max {worst:.3e} at {wi}"
          f"  uu {wuu:.3e}  status|=0x{stbits:x} nonfinite={nan}  {dt:.2f}s", flush=True)
    if worst > 1e-5 or wuu > 1e-5 or nan:
        bad += 1
print("FILES OVER 1e-5:", bad)
