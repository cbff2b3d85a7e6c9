#!/usr/bin/env python3
"""TestRuns' five examples (180 runs, command blocks of tests/golden/shipped/sbchkN.sbd) through `sbdart_amd --batch`
with SBD_TIMING=1, and through the reference one process per run (when oracle/_ref/sbdart_ref exists)."""
import gzip, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sbdart_amd.sweep import Sweep, run_directories
host = os.path.join(ROOT, "sbdart_amd", "bin", "sbdart_amd")
ref = os.path.join(ROOT, "oracle", "_ref", "sbdart_ref")
with tempfile.TemporaryDirectory() as tmp:
    dirs = []
    for n in range(1, 6):
        block = gzip.open(os.path.join(ROOT, "tests", "golden", "shipped", f"sbchk{n}.sbd.gz"), "rt").read().split("_DATA_")[0]
        s = Sweep(block)
        for it in range(len(s)):
            d = os.path.join(tmp, f"r{n}_{it:04d}")
            os.makedirs(d)
            open(os.path.join(d, "INPUT"), "w").write("\n &INPUT\n" + s.inputs(it)[0] + " /\n")
            dirs.append(d)
    lst = os.path.join(tmp, "list")
    open(lst, "w").write("\n".join(dirs) + "\n")
    for rep in range(3):
        t0 = time.perf_counter()
        p = subprocess.run([host, "--batch", lst], cwd=tmp, env=dict(os.environ, SBD_TIMING="1"), capture_output=True, text=True)
        print(f"batch: {time.perf_counter() - t0:.3f} s rc={p.returncode}", [l for l in p.stderr.splitlines() if "sbdart_amd" in l])
    if os.access(ref, os.X_OK):
        t0 = time.perf_counter()
        for d in dirs:
            subprocess.run([ref], cwd=d, capture_output=True)
        print(f"reference, one process per run, sequential: {time.perf_counter() - t0:.3f} s")
