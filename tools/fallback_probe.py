"""How many layers go to the reference-algorithm layer kernel on real record sets, and what a listed layer costs.
Run on the GPU box:  python tools/fallback_probe.py [reps]
For each golden record file: replicate its records to a batch of ~16k items, solve, report the listed-layer count and the
time per step; then a synthetic thermal-cloud batch (conservative-scattering cloud layers with a thermal source: the
class the fast kernel lists) with 0 %, 0.1 %, 1 % and 5 % of the items carrying such a layer."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
torch.cuda.init()
from sbdart_amd.engine import engine_for_record
from sbdart_amd.records import read_records

GOLDEN = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")


def run(recs, target, label, one_listed=False):
    r0 = recs[0]
    recs = [r for r in recs if r.nstr == r0.nstr and r.nlyr == r0.nlyr and np.array_equal(r.temper, r0.temper)
            and r.umu0 == r0.umu0 and (r.flags & ~1) == (r0.flags & ~1) and r.nmom == r0.nmom and len(r.umu) == len(r0.umu)]
    rep = max(1, target // len(recs))
    idx = np.tile(np.arange(len(recs)), rep)
    st_ = lambda f: np.stack([getattr(recs[i], f) for i in idx])
    ar_ = lambda f: np.array([getattr(recs[i], f) for i in idx])
    args = (st_("dtauc"), st_("ssalb"), st_("pmom"), ar_("wvnmlo"), ar_("wvnmhi"), ar_("fbeam"), ar_("albedo"),
            np.array([recs[i].plank for i in idx], dtype=np.uint8))
    if one_listed:                                        # ONE conservative thermal cloud layer in the whole batch
        args[1][len(idx) // 2, 10] = 1.0
        args[0][len(idx) // 2, 10] = 5.0
    with engine_for_record(r0, max_batch=len(idx)) as eng:
        dev = [torch.as_tensor(np.ascontiguousarray(a)).cuda() for a in args]
        dev[7] = dev[7].to(torch.uint8)
        out = eng.solve(*dev)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            out = eng.solve(*dev)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        eng.enable_timing(True)
        eng.solve(*dev)
        torch.cuda.synchronize()
        fb = eng.last_fallback_layers()
        phases = [round(eng.last_ms(k), 4) for k in range(6)]
        eng.enable_timing(False)
    t = float(np.median(ts))
    nl = len(idx) * r0.nlyr
    res = {"case": label, "items": int(len(idx)), "nstr": int(r0.nstr), "nlyr": int(r0.nlyr), "listed_layers": int(fb),
           "listed_frac_of_layers": fb / nl, "ms_per_step": 1e3 * t, "items_per_s": len(idx) / t,
           "nonzero_status": int((out[2].cpu().numpy() != 0).sum()), "phase_ms_serialized": phases}
    print(json.dumps(res), flush=True)
    return res


def synthetic(frac, n=16384, L=33, nstr=16, seed=5):
    import dataclasses
    from sbdart_amd.records import F_LAMBER, F_ONLYFL, F_PLANK, SolveRecord
    rng = np.random.default_rng(seed)
    nmom = nstr
    recs = []
    temper = np.linspace(220.0, 290.0, L + 1)
    for i in range(256):
        g = rng.uniform(0.0, 0.85, L)
        ss = rng.uniform(0.2, 0.999, L)
        dt = 10 ** rng.uniform(-3, 0.3, L)
        if rng.uniform() < frac:
            k = rng.integers(5, L - 2)
            ss[k] = 1.0; dt[k] = 5.0; g[k] = 0.85                     # a conservative cloud layer in a thermal run
        recs.append(SolveRecord(nlyr=L, nstr=nstr, nmom=nmom, flags=F_LAMBER | F_ONLYFL | F_PLANK, wvnmlo=900.0, wvnmhi=920.0,
                                fbeam=0.0, umu0=0.5, phi0=0.0, albedo=0.1, btemp=290.0, ttemp=0.0, temis=0.0,
                                dtauc=dt, ssalb=ss, temper=temper, pmom=g[:, None] ** np.arange(nmom + 1)[None, :],
                                umu=np.zeros(0), phi=np.zeros(0)))
    return recs


if __name__ == "__main__":
    out = []
    for name in ["cfgB_sw_nstr16", "cfg3_lw_nstr16_cloud", "sbchk1", "sbchk3"]:
        p = os.path.join(GOLDEN, name + ".sbdrec")
        if os.path.exists(p):
            out.append(run(list(read_records(p)), 16384, name))
    out.append(run(synthetic(0.0), 16384, "synthetic thermal, ONE listed layer in the batch", one_listed=True))
    for frac in (0.0, 0.004, 0.04, 0.3):
        out.append(run(synthetic(frac), 16384, f"synthetic thermal, {frac:g} of the items with a conservative cloud layer"))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/fallback_probe.json", "w"), indent=1)
