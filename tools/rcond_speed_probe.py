"""GPU box: what the exact errmsg-2 path costs when it is taken.  A batch of work items with an exactly conservative layer
(SSALB = 1: listed by setup_kernel's mark) at NSTR 16 x 33 and NSTR 32 x 50 layers, solved with band_rcond_kernel's wave
form and (SBD_RCOND_SERIAL=1, in a child process) its one-lane form; and the same batch with SSALB = 0.999 (nothing
listed).  Prints seconds per batch and whether the two forms return the same estimates bit for bit."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)


def run(nstr, nlyr, nitem, cons):
    import torch
    torch.cuda.init()
    from sbdart_amd.engine import DisortEngine
    rng = np.random.default_rng(5)
    k = np.arange(nstr + 3)
    g = rng.uniform(0.0, 0.85, (nitem, nlyr))
    dt = np.exp(rng.uniform(-4, 1, (nitem, nlyr)))
    w = rng.uniform(0.3, 0.999, (nitem, nlyr))
    if cons:
        w[:, nlyr // 3] = 1.0
    pm = g[:, :, None] ** k[None, None, :]
    with DisortEngine(nlyr=nlyr, nstr=nstr, nmom=nstr + 2, temper=np.linspace(220, 290, nlyr + 1), umu0=0.6, btemp=290.0, ttemp=0.0,
                      temis=0.0, onlyfl=True, level_out=[0, nlyr], device=0, max_batch=nitem) as eng:
        args = (dt, w, pm, np.full(nitem, 9000.0), np.full(nitem, 9100.0), np.ones(nitem), np.full(nitem, 0.3), np.zeros(nitem, dtype=np.uint8))
        eng.solve(*args)
        t0 = time.perf_counter()
        _, _, st = eng.solve(*args)
        dt_s = time.perf_counter() - t0
        lst = eng.debug_array(17, np.int32, 4)
        rc = eng.debug_array(16, np.float64, nitem)
    return dt_s, int(lst[0]), rc, int((np.asarray(st) != 0).sum())


if __name__ == "__main__":
    if len(sys.argv) > 1:
        nstr, nlyr, nitem, cons = (int(x) for x in sys.argv[1:5])
        s, n, rc, bad = run(nstr, nlyr, nitem, cons)
        np.save(sys.argv[5], rc)
        print(f"{s:.4f} {n} {bad}")
        sys.exit(0)
    for nstr, nlyr, nitem in ((16, 33, 2048), (32, 50, 512), (40, 65, 64)):
        res = {}
        for name, cons, env in (("nothing listed", 0, {}), ("wave form", 1, {}), ("one-lane form", 1, {"SBD_RCOND_SERIAL": "1"})):
            out = os.path.join("/tmp", f"rc_{nstr}_{name.split()[0]}.npy")
            p = subprocess.run([sys.executable, __file__, str(nstr), str(nlyr), str(nitem), str(cons), out], env=dict(os.environ, **env),
                               capture_output=True, text=True, timeout=1500)
            if p.returncode != 0:
                print(name, "failed:", p.stderr[-400:])
                continue
            s, n, bad = p.stdout.split()[-3:]
            res[name] = (float(s), int(n), np.load(out))
            print(f"NSTR {nstr} x {nlyr} layers, {nitem} items, {name}: {float(s) * 1e3:.1f} ms per batch, {n} systems listed, {bad} non-zero status", flush=True)
        if "wave form" in res and "one-lane form" in res:
            a, b = res["wave form"][2], res["one-lane form"][2]
            print("  estimates of the two forms identical:", bool(np.array_equal(a, b, equal_nan=True)), " smallest:", float(np.nanmin(a)))
