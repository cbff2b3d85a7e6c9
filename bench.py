#!/usr/bin/env python3
"""Headline benchmark: spectral-points/sec of the DISORT hot path on a synthetic
16-stream short-wave sweep (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: every (wavelength, k-term) work
item of the sweep is solved on the GPU (inputs already resident in HBM), the spectrally
integrated fluxes are accumulated with stdout1's weights (drt.f:964-1054) and, for N>1,
summed over ranks with one RCCL reduce.  Work is sharded by spectral point, weak scaling
(each rank owns `--nwl` points of a N-times longer sweep).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VEC_PEAK_TF = 78.6        # fp64 vector peak
PCIE_PEAK_GBS = 64.0           # PCIe 5.0 x16, one direction (the host entry point's bound)


def algorithmic_bytes_per_solve(nlyr, nstr, nlev):
    # in = 8*[L*(NSTR+3)+8], out = 8*3*nlev (rfldir, rfldn, flup at nlev levels)
    return 8 * (nlyr * (nstr + 3) + 8) + 8 * 3 * nlev


def algorithmic_flops_per_solve(nlyr, nstr):
    nn = nstr // 2
    ncd = 3 * nn - 1
    per_layer = 3 * nn * nstr ** 2 + 2 * nn ** 3 + 25 * nn ** 3 + 2 * nn ** 3 + (2.0 / 3.0) * nstr ** 3 + 4 * nstr ** 2
    N = nstr * nlyr
    band = 2 * N * ncd * (2 * ncd) + 2 * N * 3 * ncd
    return nlyr * per_layer + band + 2 * nstr ** 2 * 3


def sample_indices(nwork, nsample):
    """Work items strided over the WHOLE sweep (UV ... 4 um: beam-only and thermal items alike)."""
    return np.unique(np.linspace(0, nwork - 1, min(nsample, nwork)).astype(np.int64))


def cpu_baseline(sw, seconds_target=10.0):
    """Reference DISORT (oracle/_ref/disort_ref_cli, kind "reference") or the C restatement
    (kind "port") timed on the host cores over a bounded sample of the same workload: 1 core, then
    one process per core (`nproc`), each on the same sample.
    Returns (1-core dict, all-core dict or None, sample indices, reference fluxes of the sample
    [nsample][3][2]: rfldir, rfldn, flup at TOA and surface) -- the last two feed the "flux RMSE vs
    CPU" half of the metric."""
    from sbdart_amd.records import read_records, write_records
    from sbdart_amd.workload import sweep_to_records
    cli = os.path.join(ROOT, "oracle", "_ref", "disort_ref_cli")
    avg_nk = sw.nwork / sw.nwl
    ncore = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if os.path.isfile(cli) and os.access(cli, os.X_OK):
        idx = sample_indices(sw.nwork, 2000)
        nsample = len(idx)
        recs = sweep_to_records(sw, idx)
        nthermal = int(sum(bool(r.plank) for r in recs))
        with tempfile.TemporaryDirectory() as d:
            write_records(os.path.join(d, "in.sbdrec"), recs, with_out=False)
            t0 = time.time()
            out = subprocess.run([cli, "in.sbdrec", "out.sbdrec", "1"], cwd=d, capture_output=True, text=True)
            probe = time.time() - t0
            rep = max(1, min(12, int(seconds_target / max(probe, 1e-3))))
            out = subprocess.run([cli, "in.sbdrec", "out.sbdrec", str(rep)], cwd=d, capture_output=True, text=True)
            ref = None
            if os.path.exists(os.path.join(d, "out.sbdrec")):
                ro = read_records(os.path.join(d, "out.sbdrec"))
                ref = np.array([[[r.rfldir[0], r.rfldir[-1]], [r.rfldn[0], r.rfldn[-1]], [r.flup[0], r.flup[-1]]] for r in ro])
            one = None
            for line in out.stdout.splitlines():
                if line.startswith("TIMING"):
                    _, nsolve, secs = line.split()
                    sps = float(nsolve) / float(secs)
                    one = {"value": sps / avg_nk, "unit": "spectral-points/s", "cores": 1, "kind": "reference",
                           "solves_per_s": sps,
                           "sample": f"{nsample} solves strided over the whole sweep ({nthermal} thermal) x {rep} "
                                     f"repeats, reference DISORT (amdflang -O2) on 1 host core, DISORT calls only"}
            allc = None
            if one is not None and ncore > 1:
                # one process per core the scheduler gives this process, each ONE pass over a sub-sample sized from
                # the 1-core rate (the reference's 3.7 MB of memsets per call make 256 copies memory-bound: round 2's
                # leg took 565 s), and a wall cap after which the leg is dropped rather than waited for
                nsub = max(50, min(nsample, int(one["solves_per_s"] * 0.35)))
                write_records(os.path.join(d, "sub.sbdrec"), recs[:: max(1, nsample // nsub)][:nsub], with_out=False)
                nsub = len(recs[:: max(1, nsample // nsub)][:nsub])
                t0 = time.time()
                procs = [subprocess.Popen([cli, "sub.sbdrec", f"out{k}.sbdrec", "1"], cwd=d,
                                          stdout=subprocess.PIPE, text=True) for k in range(ncore)]
                secs, cap = [], 60.0
                for pr in procs:
                    try:
                        so, _ = pr.communicate(timeout=max(0.1, cap - (time.time() - t0)))
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        pr.communicate()
                        continue
                    for line in so.splitlines():
                        if line.startswith("TIMING"):
                            secs.append(float(line.split()[2]))
                if len(secs) == ncore:
                    sps = ncore * nsub / max(secs)
                    allc = {"value": sps / avg_nk, "unit": "spectral-points/s", "cores": ncore, "nproc": ncore,
                            "kind": "reference", "solves_per_s": sps, "wall_s": time.time() - t0,
                            "sample": f"{ncore} processes (sched_getaffinity) x {nsub} solves of the same sample, one "
                                      f"pass (slowest process's DISORT time), capped at {cap:.0f} s"}
                else:
                    allc = {"value": None, "cores": ncore, "kind": "reference",
                            "sample": f"dropped: {ncore - len(secs)} of {ncore} processes exceeded the {cap:.0f} s cap"}
            if one is not None:
                return one, allc, idx, ref
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle  # baseline leg only
    idx = sample_indices(sw.nwork, 1500)
    nsample = len(idx)
    recs = sweep_to_records(sw, idx)
    o0 = [pyoracle.disort(r) for r in recs]
    ref = np.array([[[o["rfldir"][0], o["rfldir"][-1]], [o["rfldn"][0], o["rfldn"][-1]], [o["flup"][0], o["flup"][-1]]] for o in o0])
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds_target:
        for r in recs:
            pyoracle.disort(r)
        n += nsample
    secs = time.time() - t0
    return {"value": n / secs / avg_nk, "unit": "spectral-points/s", "cores": 1, "kind": "port",
            "solves_per_s": n / secs,
            "sample": f"{nsample} solves strided over the whole sweep, repeated for {secs:.1f}s, C restatement "
                      f"(oracle/disort_oracle.c, gcc -O2) through ctypes on 1 host core"}, None, idx, ref


MI355X_SIMDS = 256 * 4         # CUs x SIMDs
MI355X_CLOCK_HZ = 2.4e9        # peak engine clock (MI355X_MICROARCH.md)
VALU_SLOT_NS_MEASURED = 2.05   # one wave64 issue slot (4 cycles) as the chip sustains it under fp64 load: 1.95 GHz
                               # (tools/microbench/valu_rates.hip, profiles/r05_valu_rates.txt)


PROFILE_TAG = "r06"


def load_profile(kind, nstr, nlyr, shape=""):
    """profiles/<tag>[_<shape>]_<kind>.json (kind: "valu" | "traffic") -- counters recorded in their own rocprofv3 --pmc
    passes (tools/profile_round.sh), NOT observations of this run.  Used only when recorded for THESE kernel sources
    (kernel_source_hash) and this shape; otherwise (None, why) -- a stale profile must not dress up a new kernel."""
    from sbdart_amd._srchash import kernel_source_hash
    name = f"{PROFILE_TAG}{'_' + shape if shape else ''}_{kind}.json"
    path = os.path.join(ROOT, "profiles", name)
    try:
        d = json.load(open(path))
    except Exception:
        return None, f"profiles/{name} missing"
    if d.get("nstr") != nstr or d.get("nlyr") != nlyr:
        return None, f"profiles/{name} is for another shape"
    if d.get("kernel_source_hash") != kernel_source_hash():
        return None, (f"profiles/{name} was recorded for kernel sources {d.get('kernel_source_hash')}, "
                      f"these are {kernel_source_hash()}: stale, not used")
    return d, f"profiles/{name} (kernel sources {d['kernel_source_hash']})"


def valu_issue(W, step_s, nstr, nlyr, shape=""):
    """Executed-instruction occupancy of the vector ALUs over the timed step: VALU wave-instructions per solve
    (rocprofv3 --pmc SQ_INSTS_VALU of this command, committed under profiles/ like the traffic figures) x the
    step's solves x 4 issue cycles per wave64 instruction / (1 024 SIMDs x 2.4 GHz x the step's time)."""
    vp, src = load_profile("valu", nstr, nlyr, shape)
    if vp is None:
        return {"valu_wave_insts_per_solve": None, "source": src}
    per_solve = sum(k["valu_wave_insts_per_solve"] for k in vp["kernels"].values())
    frac = per_solve * W * 4.0 / (MI355X_SIMDS * MI355X_CLOCK_HZ * step_s)
    return {"valu_wave_insts_per_solve": per_solve, "issue_cycles_per_inst": 4,
            "frac_of_step": frac,
            "frac_of_step_at_sustained_clock": per_solve * W * VALU_SLOT_NS_MEASURED * 1e-9 / (MI355X_SIMDS * step_s),
            "sustained_slot_ns": VALU_SLOT_NS_MEASURED, "source": src + " (SQ_INSTS_VALU, separate PMC pass)",
            "per_kernel": {k: v["valu_wave_insts_per_solve"] for k, v in vp["kernels"].items()}}


def rocprof_kernel_name(phase, nstr, fused, rad=False):
    """The name a rocprofv3 kernel trace prints for the kernel family of timing phase `phase` at this stream count (the
    template arguments are the ones sbd_launch.hpp instantiates): so that `roofline.frac` can be recomputed from
    profiles/<tag>_kernel_stats.csv mechanically."""
    nn = nstr // 2
    g = 1
    while g < nn:
        g *= 2
    if nn > 16:
        g = nn                                   # NSTR 34-40: groups of nn lanes (sbd_layer2.hpp)
    b = "true" if fused else "false"
    if phase == "setup_kernel":
        return "sbd::setup_kernel(sbd::Params)"
    if phase == "layer_kernel":
        return f"void sbd::layer_kernel2<{nn}, {g}, {'true' if rad else 'false'}>(sbd::Params, int*)"
    if phase == "band_kernel":
        if nstr <= 16:
            return f"void sbd::band4_kernel<{nn}, {b}, false, false>(sbd::Params)"
        if nstr <= 32:
            return f"void sbd::band1_kernel<{nn}, {b}>(sbd::Params)"
        return f"void sbd::band_rows_kernel<{nn}, {b}>(sbd::Params)"
    if phase == "backsolve_kernel":
        if nstr <= 16:
            return f"void sbd::backsolve4_kernel<{nn}>(sbd::Params)"
        return f"void sbd::backsolve1_kernel<{nn}>(sbd::Params)" if nstr <= 32 else f"void sbd::backsolve_kernel<{nn}>(sbd::Params)"
    return "sbd::usrint_kernel(sbd::Params) + sbd::azimuth_kernel(sbd::Params, int)"


def shape_roofline(names, phase_ms, nlaunch, pass_size, W, nstr, nlyr, nlev, shape="", extra_out_bytes=0, fused=True, rad=False):
    """The `roofline` object for the kernel that takes the most time: algorithmic bytes over its time."""
    dom = int(np.argmax(phase_ms))
    abytes = algorithmic_bytes_per_solve(nlyr, nstr, nlev) + extra_out_bytes
    ach = abytes * W / (phase_ms[dom] * 1e-3) / 1e9
    traffic = None
    tp, src = load_profile("traffic", nstr, nlyr, shape)
    if tp is not None and tp.get("solves_per_launch") == pass_size:
        key = {"layer_kernel": "layer_kernel2", "band_kernel": ("band4_kernel", "band1_kernel", "band_rows_kernel", "band_kernel")}.get(names[dom], names[dom])
        keys = key if isinstance(key, tuple) else (key,)
        for kname, kd in tp["kernels"].items():
            if any(k in kname for k in keys):
                traffic = (traffic or 0.0) + kd["bytes_per_launch"]
    elif tp is not None:
        src = f"{src}: recorded at {tp.get('solves_per_launch')} solves per launch, this run has {pass_size}"
    return {"bound": "hbm", "kernel": rocprof_kernel_name(names[dom], nstr, fused, rad), "kernel_family": names[dom], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_unit": f"HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE; {src})",
            "algorithmic_bytes_per_solve": abytes, "solves_per_launch": pass_size, "launches": nlaunch,
            "avg_launch_ms": float(phase_ms[dom] / max(1, nlaunch))}


def latency_case(device):
    """The drop-in case at its real size: BASELINE configs[1] (0.25-4.0 um, wlinc 0.005, NSTR 16: 751 wavelengths,
    2 009 solves).  The Fortran host's band model writes the work items (SBD_DUMP_OPTICS stops it before the
    engine), the host entry point sbd_fleet_solve_host solves them from pageable arrays -- create excluded."""
    from sbdart_amd.engine import DisortFleet
    from sbdart_amd.records import read_records
    host = os.path.join(ROOT, "sbdart_amd", "bin", "sbdart_amd")
    if not os.path.exists(host):
        return None
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "INPUT"), "w") as f:
            f.write("\n &INPUT\n idatm=6 isat=0 wlinf=.25 wlsup=4.0 wlinc=.005 nstr=16 iout=10\n /\n")
        out = os.path.join(d, "items.sbdrec")
        subprocess.run([host], cwd=d, env=dict(os.environ, SBD_DUMP_OPTICS=out, SBD_OPTICS=os.path.join(d, "none")),
                       capture_output=True, text=True)
        if not os.path.exists(out):
            return None
        recs = [r for r in read_records(out) if r.ff != 0.0]
    r0 = recs[0]
    nwl = len({r.iwl for r in recs})
    arrs = [np.stack([r.dtauc for r in recs]), np.stack([r.ssalb for r in recs]), np.stack([r.pmom for r in recs]),
            np.array([r.wvnmlo for r in recs]), np.array([r.wvnmhi for r in recs]), np.array([r.fbeam for r in recs]),
            np.array([r.albedo for r in recs]), np.array([r.plank for r in recs], dtype=np.uint8)]
    w = np.array([r.wt * r.ff for r in recs])
    t0 = time.perf_counter()
    fleet = DisortFleet(nlyr=r0.nlyr, nstr=r0.nstr, nmom=r0.nmom, temper=r0.temper, umu0=r0.umu0, btemp=r0.btemp,
                        ttemp=r0.ttemp, temis=r0.temis, onlyfl=True, level_out=[0, r0.nlyr], devices=[device],
                        max_batch=len(recs))
    create_s = time.perf_counter() - t0
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        res = fleet.solve(*arrs, weight=w, items=False)
        ts.append(time.perf_counter() - t0)
    st = res[2]
    fleet.close()
    ts = sorted(ts[1:])
    med = ts[len(ts) // 2]
    return {"workload": "BASELINE configs[1] from INPUT: band model (Fortran host) -> sbd_fleet_solve_host, pageable arrays",
            "nwl": nwl, "solves": len(recs), "ms_median": 1e3 * med, "ms_min": 1e3 * ts[0], "ms_max": 1e3 * ts[-1],
            "spectral_points_per_s": nwl / med, "create_s_excluded": create_s,
            "nonzero_status": int(np.count_nonzero(st)), "repeats": len(ts)}


def e2e_input_to_stdout(device):
    """INPUT -> stdout of the drop-in executable (sbdart_amd/bin/sbdart_amd: namelist, band model, engine, writers) on
    BASELINE's sweeps, as a caller of `sbdart` sees it: wall time of the process (median of 3 after one discarded run) and
    the host's own account (SBD_TIMING: seconds inside the process from its first statement) -- `points_per_s_in_process`
    excludes exec / dynamic loading / teardown, `points_per_s_wall` does not.  Never `value`."""
    host = os.path.join(ROOT, "sbdart_amd", "bin", "sbdart_amd")
    if not os.path.exists(host):
        return None
    cases = (("configs[1] wlinc=.005", "idatm=6 isat=0 wlinf=.25 wlsup=4.0 wlinc=.005 nstr=16 iout=10"),
             ("configs[1] wlinc=.00005", "idatm=6 isat=0 wlinf=.25 wlsup=4.0 wlinc=.00005 nstr=16 iout=10"),
             ("configs[4] 0.25-100um, 1 cm-1 steps, nstr=32, 50 layers", "idatm=6 isat=0 wlinf=.25 wlsup=100 wlinc=1.0001 nstr=32 ngrid=50 iout=10 sza=30"))
    out = {}
    env = dict(os.environ, SBD_TIMING="1", SBD_OPTICS="/nonexistent", SBD_DEVICES=str(device))
    for name, nml in cases:
        try:
            with tempfile.TemporaryDirectory() as d:
                with open(os.path.join(d, "INPUT"), "w") as f:
                    f.write(f"\n &INPUT\n {nml}\n /\n")
                walls, acct, text = [], {}, ""
                for rep in range(4):
                    t0 = time.perf_counter()
                    r = subprocess.run([host], cwd=d, env=env, capture_output=True, text=True, timeout=600)
                    w = time.perf_counter() - t0
                    if r.returncode != 0:
                        raise RuntimeError(f"exit code {r.returncode}: {r.stderr[-300:]}")
                    if rep:
                        walls.append(w)
                    text = r.stdout
                    for line in r.stderr.splitlines():
                        if line.startswith("sbdart_amd: timing "):
                            kv = dict(x.split("=") for x in line.split()[2:])
                            if rep:
                                for k, v in kv.items():
                                    acct.setdefault(k, []).append(float(v))
                walls.sort()
                med = lambda v: float(sorted(v)[len(v) // 2])
                a = {k: med(v) for k, v in acct.items()}
                nwl = int(a.get("nwl", 0))
                rec = {"namelist": nml, "wall_s_median": walls[len(walls) // 2], "wall_s_min": walls[0], "wall_s_max": walls[-1],
                       "account_s": a, "stdout_tokens": len(text.split())}
                if nwl:
                    rec["nwl"] = nwl
                    rec["points_per_s_wall"] = nwl / rec["wall_s_median"]
                    if a.get("total"):
                        rec["points_per_s_in_process"] = nwl / a["total"]
                out[name] = rec
        except Exception as ex:   # a side line must not take the headline down
            out[name] = {"error": repr(ex)}
    try:
        out["served"] = served_runs(device, cases)
    except Exception as ex:
        out["served"] = {"error": repr(ex)}
    return out


def served_runs(device, cases):
    """The same INPUTs as a caller of `sbdart` sees them when `sbdart` is the CLIENT of a resident `sbdart_amd --serve`
    (sbdart_amd/fortran/sbdart_client.c; the harnesses of the reference launch one `sbdart` per run, RunRT.py:2021-2044):
    wall time of the client process, median of 5 after one discarded run, the text compared with the one-shot process's.
    `testruns_180`: TestRuns' five examples, 180 runs launched one client process per run, sequentially, as
    TestRuns/test_runs launches the reference (which needs ~0.8 s for them on this box's host, `reference_s`)."""
    import gzip
    bindir = os.path.join(ROOT, "sbdart_amd", "bin")
    host, client = os.path.join(bindir, "sbdart_amd"), os.path.join(bindir, "sbdart")
    if not os.path.exists(client):
        return None
    out = {}
    with tempfile.TemporaryDirectory() as top:
        sock = os.path.join(top, "sock")
        senv = dict(os.environ, SBD_OPTICS="/nonexistent", SBD_DEVICES=str(device), SBDART_AMD_IDLE_S="120")
        srv = subprocess.Popen([host, "--serve", sock], env=senv, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        try:
            for _ in range(6000):
                if os.path.exists(sock) or srv.poll() is not None:
                    break
                time.sleep(0.01)
            cenv = dict(os.environ, SBDART_AMD_SOCKET=sock, SBDART_AMD_NO_AUTOSTART="1")
            for k, (name, nml) in enumerate(cases):
                d = os.path.join(top, f"case{k}")
                os.makedirs(d)
                with open(os.path.join(d, "INPUT"), "w") as f:
                    f.write(f"\n &INPUT\n {nml}\n /\n")
                alone = subprocess.run([host], cwd=d, env=senv, capture_output=True, text=True, timeout=600).stdout
                walls, same = [], True
                for rep in range(6):
                    t0 = time.perf_counter()
                    r = subprocess.run([client], cwd=d, env=cenv, capture_output=True, text=True, timeout=600)
                    w = time.perf_counter() - t0
                    same = same and r.returncode == 0 and r.stdout == alone
                    if rep:
                        walls.append(w)
                walls.sort()
                out[name] = {"wall_s_median": walls[len(walls) // 2], "wall_s_min": walls[0], "wall_s_max": walls[-1],
                             "text_equals_one_shot": bool(same)}
            # TestRuns' 180 runs, one client process per run
            from sbdart_amd.sweep import Sweep
            shipped = os.path.join(ROOT, "tests", "golden", "shipped")
            nrun, secs = 0, 0.0
            for k in range(1, 6):
                block = gzip.open(os.path.join(shipped, f"sbchk{k}.sbd.gz"), "rt").read().split("_DATA_", 1)[0]
                sw = Sweep(block)
                t0 = time.perf_counter()
                outs = sw.run(client, os.path.join(top, f"tr{k}"), env=cenv)
                secs += time.perf_counter() - t0
                nrun += len(outs)
            out["testruns_180"] = {"runs": nrun, "seconds": secs, "ms_per_run": 1e3 * secs / max(1, nrun),
                                   "how": "one `sbdart` client process per run, sequentially (subprocess.run per run, INPUT written before it)"}
            ref = os.path.join(ROOT, "oracle", "_ref", "sbdart_ref")
            if os.path.exists(ref):
                t0 = time.perf_counter()
                for k in range(1, 6):
                    block = gzip.open(os.path.join(shipped, f"sbchk{k}.sbd.gz"), "rt").read().split("_DATA_", 1)[0]
                    Sweep(block).run(ref, os.path.join(top, f"rr{k}"))
                out["testruns_180"]["reference_s"] = time.perf_counter() - t0
        finally:
            srv.terminate()
            try:
                srv.wait(timeout=10)
            except subprocess.TimeoutExpired:
                srv.kill()
    return out


def other_shapes(dev, only=None, serialized_pass=True):
    """One-step lines for the other BASELINE shapes (parity-test cases, not the headline): configs[4]'s
    NSTR 32 x 50 layers in flux mode, configs[3]'s radiance shape (NSTR 32, 20 x 16 angles, 32 azimuth modes), and the
    reference's largest stream count (NSTR 40, params.f:9-11) on the headline's 33 layers.
    `only` = "cfgC" | "cfgD" | "nstr40": that shape alone (bench.py --shape: the command the shape's counters are recorded with)."""
    import torch
    from sbdart_amd.engine import DisortEngine
    from sbdart_amd.workload import sw_sweep
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = {}
    for name, kw, nwl in (("cfgD_nstr32_50layers_flux", dict(nstr=32, nlyr=50), 6144),
                          ("cfgC_nstr32_radiance_20x16", dict(nstr=32, nlyr=33, thermal_above_um=99.0), 384),
                          ("nstr40_33layers_flux", dict(nstr=40, nlyr=33), 4096)):     # params.f's largest stream count
        if only and not name.startswith(only):
            continue
        try:
            sw = sw_sweep(nwl=nwl, seed=12345, **kw)
            rad = "radiance" in name
            ekw = dict(onlyfl=True)
            if rad:
                ekw = dict(onlyfl=False, umu=np.cos(np.deg2rad(np.linspace(0, 85, 20)[::-1])), phi=np.linspace(0, 180, 16))
            eng = DisortEngine(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp,
                               ttemp=sw.ttemp, temis=0.0 if rad else sw.temis, level_out=[0, sw.nlyr],
                               device=dev.index or 0, **ekw)
            ins = [t(sw.dtauc), t(sw.ssalb), t(sw.pmom), t(sw.wvnmlo), t(sw.wvnmhi), t(sw.fbeam), t(sw.albedo), t(sw.plank)]
            eng.solve(*ins)
            torch.cuda.synchronize()
            eng._L.sbd_engine_enable_timing(eng._h, 2)     # events where the passes run, read after the timed steps
            t0 = time.perf_counter()
            nrep = 2
            for _ in range(nrep):
                flux, uu, st = eng.solve(*ins)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / nrep
            kms_ip = np.array([eng.last_ms(p) for p in range(5)])
            if serialized_pass:                       # each kernel family alone on the chip (one stream, a sync per pass)
                eng.enable_timing(True)
                eng.solve(*ins)
                torch.cuda.synchronize()
                kms = [eng.last_ms(p) for p in range(5)]
                fb = eng.last_fallback_layers()
            else:
                kms, fb = [float(x) for x in kms_ip], -1
            finite = bool(torch.isfinite(flux).all().item()) and (uu is None or bool(torch.isfinite(uu).all().item()))
            names = ["setup_kernel", "layer_kernel", "band_kernel", "backsolve_kernel", "usrint+azimuth"]
            nl = eng.pass_count(sw.nwork)
            shape = "cfgC" if rad else ("nstr40" if name.startswith("nstr40") else "cfgD")
            out[name] = {"value": sw.nwl / dt if finite else None, "unit": "spectral-points/s", "ms_per_step": 1e3 * dt,
                         "nwl": sw.nwl, "solves": sw.nwork, "nstr": sw.nstr, "nlyr": sw.nlyr,
                         "kernel_ms": dict(zip(["setup", "layer", "band", "backsolve", "usrint+azimuth"], map(float, kms))),
                         "nonzero_status": int((st != 0).sum().item()), "fallback_layers": int(fb), "finite": finite,
                         "kernel_ms_in_timed_step": dict(zip(["setup", "layer", "band", "backsolve", "usrint+azimuth"], map(float, kms_ip))),
                         "roofline": shape_roofline(names, kms_ip if (kms_ip > 0).all() else np.array(kms), nl, (sw.nwork + nl - 1) // nl, sw.nwork, sw.nstr,
                                                    sw.nlyr, 2, shape, extra_out_bytes=8 * 20 * 16 if rad else 0, fused=not rad, rad=rad),
                         "valu_issue": valu_issue(sw.nwork, dt, sw.nstr, sw.nlyr, shape)}
            eng.close()
        except Exception as ex:   # a side line must not take the headline down
            out[name] = {"value": None, "error": repr(ex)}
    return out


def host_entry_legs(args, sw, eng, d_in, d_w, acc, flux, status, stream, dev, local_rank, rank, world, level_out, barrier, strong, coll):
    """The step through the HOST entry points (never `value`): DISORT's arguments as arrays, the moments once per spectral
    point, and the compact form of SURVEY 8(d)'s engine phase (`value_8d`).  Returns the fields of the bench line."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    W = sw.nwork
    L = eng._L
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # ---- the same step through the HOST entry point (what the Fortran host calls): inputs in pinned host
    #      memory, every pass stages its slice H2D on its stream beside the other pass's kernels, weighted
    #      sums on the device, D2H of the sums and the status words (SURVEY 8d's "engine phase"); never `value` ----
    from sbdart_amd.engine import DisortFleet
    fleet = DisortFleet(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp,
                        ttemp=sw.ttemp, temis=sw.temis, onlyfl=True, level_out=level_out, devices=[local_rank])
    h_in = [x.cpu().pin_memory().numpy() for x in d_in]
    h_w = d_w.cpu().pin_memory().numpy()

    def step_host():
        return fleet.solve(*h_in, weight=h_w, items=False)[3]

    acc_h = step_host()
    barrier()
    t0 = time.perf_counter()
    nh = max(1, min(args.steps, 5))
    for _ in range(nh):
        acc_h = step_host()
    barrier()
    elapsed_h = time.perf_counter() - t0
    # ... and as the drop-in host delivers a sweep: the phase-function moments once per SPECTRAL POINT, shared by the
    # point's k-terms (sbd_batch_in::pmom_row; drt.f:476-533 computes them before the k loop).  The synthetic sweep draws
    # its moments per work item, so this variant gives every k-term its point's first set: the same sizes and kernels,
    # 2.3 x fewer bytes over PCIe.  A side number (`value_incl_h2d_shared_moments`), never `value`.
    rows = np.ascontiguousarray(sw.wl_of, dtype=np.int32)
    first = np.concatenate([[0], np.nonzero(np.diff(rows))[0] + 1])
    h_pm_pt = torch.from_numpy(np.ascontiguousarray(sw.pmom[first])).pin_memory().numpy()
    h_rows = torch.from_numpy(rows).pin_memory().numpy()

    def step_host_shared():
        return fleet.solve(h_in[0], h_in[1], h_pm_pt, *h_in[3:], weight=h_w, items=False, pmom_row=h_rows)[3]

    step_host_shared()
    barrier()
    t0 = time.perf_counter()
    for _ in range(nh):
        step_host_shared()
    barrier()
    elapsed_hs = time.perf_counter() - t0
    # ... and from the COMPACT form of a batch (sbd_mix_in: per spectral point the scatterers, per item the gas of its
    # k-term; DTAUC / SSALB / PMOM assembled on the device): SURVEY 8(d)'s engine phase -- H2D + kernels + D2H/reduce --
    # with the H2D a ninth of the bytes.  The same sizes and k-term structure as the headline sweep, its optical
    # properties drawn per spectral point (sbdart_amd/workload.py: sw_sweep_mix); `value_8d`.  The same sweep's rate
    # with its assembled arrays resident in HBM is measured beside it (value_8d_resident): the ratio is the cost of
    # feeding the engine from the host.
    from sbdart_amd.workload import sw_sweep_mix
    if strong:
        from sbdart_amd.shard import shard_range
        p_lo, p_hi = shard_range(args.nwl, rank, world)
        mx = sw_sweep_mix(nwl=args.nwl, nstr=args.nstr, nlyr=args.nlyr, seed=12345, shard=0).points(p_lo, p_hi)
    else:
        mx = sw_sweep_mix(nwl=args.nwl, nstr=args.nstr, nlyr=args.nlyr, seed=12345, shard=rank)
    pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()
    m_in = [x if isinstance(x, tuple) else pin(x) for x in mx.mix_args()]
    m_w = pin(mx.weight)

    def step_mix():
        return fleet.solve_mix(*m_in, weight=m_w, items=False)[3]

    acc_m = step_mix()
    barrier()
    t0 = time.perf_counter()
    mix_steps = []
    for _ in range(nh):
        t1 = time.perf_counter()
        acc_m = step_mix()
        mix_steps.append(1e3 * (time.perf_counter() - t1))
    barrier()
    elapsed_m = time.perf_counter() - t0
    fleet.close()
    # the same sweep, arrays resident
    md, msa, mp = mx.arrays()
    r_in = [t(md), t(msa), t(mp), t(mx.wvnmlo[mx.point_of]), t(mx.wvnmhi[mx.point_of]), t(mx.fbeam[mx.point_of]),
            t(mx.albedo[mx.point_of]), t(mx.plank[mx.point_of])]
    r_row = t(mx.point_of.astype(np.int32))
    r_w = t(mx.weight)
    acc_r = torch.zeros((5, eng.nlev), dtype=torch.float64, device=dev)
    flux_r = torch.empty_like(flux)                     # (its own outputs: `flux` / `status` belong to the headline sweep)
    status_r = torch.empty_like(status)

    def step_res():
        eng.solve_device(*r_in, out=(flux_r, None, status_r), stream=stream, pmom_row=r_row)
        acc_r.zero_()
        rc = L.sbd_engine_accumulate_device(eng._h, W, r_w.data_ptr(), flux_r.data_ptr(), None, acc_r.data_ptr(), None, C.c_void_p(stream))
        assert rc == 0, rc

    step_res()
    barrier()
    t0 = time.perf_counter()
    for _ in range(nh):
        step_res()
    barrier()
    elapsed_r = time.perf_counter() - t0
    mix_agree = bool(np.allclose(acc_m, acc_r.cpu().numpy(), rtol=1e-12, atol=0))
    bad_mix = int((status_r != 0).sum().item())
    del r_in, r_row, r_w, flux_r, status_r
    elapsed_m, elapsed_r = coll.all_max(elapsed_m, elapsed_r)
    h2d_bytes = int(sum(a.nbytes for a in h_in) + h_w.nbytes)
    h2d_bytes_shared = int(h2d_bytes - h_in[2].nbytes + h_pm_pt.nbytes + h_rows.nbytes)
    elapsed_h, elapsed_hs = coll.all_max(elapsed_h, elapsed_hs)

    assert np.allclose(acc_h, acc.cpu().numpy(), rtol=1e-12, atol=0) or world > 1, "host entry point disagrees with the device one"
    nwl_total = args.nwl if strong else sw.nwl * world
    med_m = float(np.median(mix_steps))
    # (every rank's own median, the slowest rank's is the node's pace -- and the same statistic on both
    #  sides of the resident ratio)
    med_m = coll.all_max(med_m)
    med_r = 1e3 * elapsed_r / nh
    return {
        "value_incl_h2d": nwl_total * nh / elapsed_h, "ms_per_step_incl_h2d": 1e3 * elapsed_h / nh,
        "value_incl_h2d_shared_moments": nwl_total * nh / elapsed_hs, "ms_per_step_incl_h2d_shared_moments": 1e3 * elapsed_hs / nh,
        # SURVEY 8(d)'s own wording of the metric ("H2D of inputs + kernels + D2H/reduce"): the host entry point's
        # rate; its bound is the PCIe link, not HBM (roofline_pcie).  `value` stays the HBM-resident rate.
        # (median of the steps, MAX over the ranks: one step in five runs takes 70 ms instead of 9 on some boxes of the pool --
        #  a stall of the host side; rank 0's steps are listed in ms_steps_8d, the mean is ms_per_step_8d_mean)
        "value_8d": nwl_total / (1e-3 * med_m), "ms_per_step_8d": med_m,
        "ms_per_step_8d_mean": 1e3 * elapsed_m / nh,
        "ms_steps_8d": [round(x, 2) for x in mix_steps],
        "value_8d_resident": nwl_total * nh / elapsed_r, "ms_per_step_8d_resident": med_r,
        "value_8d_over_resident": (1e3 * elapsed_r / nh) / (1e3 * elapsed_m / nh),
        "value_8d_note": "SURVEY 8(d)'s engine phase (H2D + kernels + D2H/reduce) through sbd_fleet_solve_mix_host: the batch "
                         "in compact form (per spectral point the scatterers, per item the gas of its k-term), DTAUC / SSALB / "
                         "PMOM assembled on the device; value_8d_resident = the same sweep with the assembled arrays already in "
                         f"HBM (value_8d_over_resident: mean step over mean step); sums of the two agree to 1e-12: {mix_agree}; "
                         f"nonzero status {bad_mix}",
        "roofline_pcie": {"bound": "pcie", "achieved": h2d_bytes * world * nh / elapsed_h / 1e9, "peak": PCIE_PEAK_GBS,
                          "unit": "GB/s", "frac": h2d_bytes * nh / elapsed_h / 1e9 / PCIE_PEAK_GBS,
                          "bytes_per_step": h2d_bytes, "bytes_per_step_shared_moments": h2d_bytes_shared,
                          "bytes_per_step_compact": mx.h2d_bytes(),
                          "achieved_shared_moments": h2d_bytes_shared * nh / elapsed_hs / 1e9,
                          "achieved_compact": mx.h2d_bytes() * nh / elapsed_m / 1e9,
                          "note": "H2D bytes of one step's inputs (per-item moments / moments once per spectral "
                                  "point / compact form) over the host-entry-point step time; peak = PCIe 5.0 x16 per "
                                  "direction.  `achieved`/`frac` belong to value_incl_h2d (DISORT's arguments as arrays: "
                                  "PCIe-bound); the compact form needs a ninth of the bytes and is compute-bound again"},
        "incl_h2d_note": "same step through the host entry point (sbd_fleet_solve_host): inputs in pinned host memory, the passes' H2D back to back on a copy stream beside the kernels, sums on the device, D2H of sums + status",
    }


DEV_SWITCHES = ("SBD_CHUNK", "SBD_WORKSPACE_MB", "SBD_BAND_V1", "SBD_LAYER_V1",
                "SBD_FORCE_EIG_FALLBACK", "SBD_DEBUG_SYNC", "SBD_DBG_FLAGS", "SBD_NO_FUSE", "SBD_SOLVE_V1", "SBD_NO_HINT", "SBD_EXACT_PIVOT", "SBD_RCOND_SERIAL")


def default_nwl(scaling, gpus):
    """--nwl when it is not given: 49 152 spectral points per GPU -- per rank with weak scaling, and with strong scaling ONE
    sweep of 49 152 x gpus points, so that every rank's shard is still a bench-size batch (2^17 solves: >= 64 work items
    per CU) and a strong-scaling run measures kernels, not launch latency."""
    return 49152 * (gpus if scaling == "strong" else 1)


class Coll:
    """The collectives of the bench line.  world > 1 on distinct GPUs: RCCL on device tensors (backend "nccl").
    --share-device rehearsal: the same calls on host copies over gloo (several ranks on one GPU cannot form an RCCL
    communicator)."""
    def __init__(self, world, dev, host):
        self.world, self.dev, self.host = world, dev, host

    def _all(self, vals, op):
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return [float(v) for v in vals]
        tt = torch.tensor([float(v) for v in vals], dtype=torch.float64, device="cpu" if self.host else self.dev)
        dist.all_reduce(tt, op=op)
        return [float(x) for x in tt.tolist()]

    def all_max(self, *vals):
        import torch.distributed as dist
        r = self._all(vals, dist.ReduceOp.MAX)
        return r[0] if len(r) == 1 else r

    def all_sum(self, *vals):
        import torch.distributed as dist
        r = self._all(vals, dist.ReduceOp.SUM)
        return r[0] if len(r) == 1 else r

    def gather(self, rank, vals):
        """[world][len(vals)] of every rank's numbers (an all-reduce of a one-hot block: no object collectives)."""
        import torch.distributed as dist
        k = len(vals)
        v = [0.0] * (k * self.world)
        v[rank * k:(rank + 1) * k] = [float(x) for x in vals]
        r = self._all(v, dist.ReduceOp.SUM) if self.world > 1 else v
        return [r[i * k:(i + 1) * k] for i in range(self.world)]

    def reduce_sum_(self, acc):
        """acc (device tensor) := sum over the ranks, on rank 0 -- the ONE collective of the path."""
        import torch.distributed as dist
        if self.world == 1:
            return
        if self.host:
            h = acc.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            acc.copy_(h)
        else:
            dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)

    def barrier(self):
        import torch.distributed as dist
        if self.world > 1:
            dist.barrier()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nwl", type=int, default=None,
                    help="spectral points: per GPU with --scaling weak (default 49152; W = 2.67x), of the ONE sweep with "
                         "--scaling strong (default 49152 x --gpus, so that every rank's shard is a bench-size batch and a "
                         "strong-scaling run measures kernels, not launches)")
    ap.add_argument("--nstr", type=int, default=16)
    ap.add_argument("--nlyr", type=int, default=33)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0, help="CPU work of the 1-core cpu_baseline leg (bounded sample)")
    ap.add_argument("--no-side-lines", action="store_true", help="skip the latency case and the other BASELINE shapes")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --nwl spectral points PER GPU (the default; each rank its own PRNG stream); strong: ONE sweep of "
                         "--nwl points cut between spectral points by sbd_shard_range, rank r solves its shard (north_star's split)")
    ap.add_argument("--headline-only", action="store_true",
                    help="the timed steps and the per-kernel timing pass only: no host-entry-point legs (whose passes have other "
                         "sizes) -- the command profiles/<tag>_kernel_stats.csv is recorded with, so that its averages are "
                         "bench-size launches")
    ap.add_argument("--shape", choices=["cfgC", "cfgD", "nstr40"], default=None,
                    help="print the one-step line of another BASELINE shape (other_shapes) and nothing else: the command its "
                         "counters in profiles/ are recorded with")
    ap.add_argument("--share-device", action="store_true",
                    help="DEVELOPER / TEST ONLY (refused unless SBD_BENCH_SHARE_DEVICE_TEST=1 is in the environment): all "
                         "--gpus ranks run on cuda:0 and meet over gloo -- a rehearsal of the N > 1 line on a one-GPU box "
                         "(tests/test_bench_rehearsal.py); the line it prints carries \"rehearsal\": true and is NOT a "
                         "multi-GPU measurement")
    ap.add_argument("--rendezvous-only", choices=["nccl", "gloo"], default=None,
                    help="launcher check: bring up --gpus ranks on this backend, count them, print that, exit (no bench line)")
    args = ap.parse_args()
    if args.share_device and os.environ.get("SBD_BENCH_SHARE_DEVICE_TEST") != "1":
        sys.exit("bench.py: --share-device is a test rehearsal (several ranks on ONE GPU); refused without "
                 "SBD_BENCH_SHARE_DEVICE_TEST=1 -- it never produces a multi-GPU measurement")
    if args.nwl is None:
        args.nwl = default_nwl(args.scaling, args.gpus)
    on = [k for k in DEV_SWITCHES if os.environ.get(k)]
    if on:   # the headline number is the default path only
        sys.exit(f"bench.py: developer switch(es) {on} set in the environment -- refusing to produce a bench line")

    import torch
    import torch.distributed as dist
    from sbdart_amd.launch import launched_world, relaunch_one_rank_per_gpu, rendezvous
    if args.gpus > 1 and launched_world() is None:
        # `python bench.py --gpus N` on its own: this process becomes the launcher of N ranks (one per GPU) and
        # passes their exit code on -- it never prints a bench line itself (a 1-GPU line labelled N would be a lie)
        backend = "gloo" if (args.rendezvous_only == "gloo" or args.share_device) else "nccl"
        if backend == "nccl" and torch.cuda.device_count() < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but this node shows {torch.cuda.device_count()} GPU(s): "
                     f"refusing to print a line for fewer devices than asked for")
        sys.exit(relaunch_one_rank_per_gpu(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    rank, local_rank, world = rendezvous(args.gpus, backend="gloo" if (args.rendezvous_only == "gloo" or args.share_device) else "nccl")
    share = bool(args.share_device)
    if share:
        local_rank = 0                     # (rehearsal: every rank on cuda:0; collectives on host tensors over gloo)
    if args.rendezvous_only:
        # launcher path only (CPU test of `--gpus N`): the ranks met, counted each other, rank 0 says so -- no metric
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world, "backend": args.rendezvous_only,
                              "ranks_counted": world}))
        if world > 1:
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll = Coll(world, dev, host=share)

    from sbdart_amd.engine import DisortEngine
    from sbdart_amd.workload import sw_sweep

    if args.shape:
        # a side shape on its own (rank 0's device): the line its counters in profiles/ are recorded for
        if rank == 0:
            print(json.dumps({"shape": args.shape, **other_shapes(dev, only=args.shape, serialized_pass=not args.headline_only)}))
        if world > 1:
            dist.destroy_process_group()
        return
    strong = args.scaling == "strong"
    p_lo, p_hi = 0, args.nwl
    if strong:
        # ONE sweep of --nwl spectral points, the same on every rank (same PRNG stream); rank r takes the points
        # sbd_shard_range gives it -- shards cut between spectral points, total work fixed as N grows
        from sbdart_amd.shard import shard_range
        p_lo, p_hi = shard_range(args.nwl, rank, world)
        sw = sw_sweep(nwl=args.nwl, nstr=args.nstr, nlyr=args.nlyr, seed=12345, shard=0).points(p_lo, p_hi)
    else:
        sw = sw_sweep(nwl=args.nwl, nstr=args.nstr, nlyr=args.nlyr, seed=12345, shard=rank)
    W = sw.nwork
    level_out = [0, sw.nlyr]                      # ntop, nbot of IOUT 1/10 (drt.f:376-381)
    eng = DisortEngine(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0,
                       btemp=sw.btemp, ttemp=sw.ttemp, temis=sw.temis, onlyfl=True,
                       level_out=level_out, device=local_rank)
    npass_headline = eng.pass_count(W)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_in = [t(sw.dtauc), t(sw.ssalb), t(sw.pmom), t(sw.wvnmlo), t(sw.wvnmhi), t(sw.fbeam), t(sw.albedo), t(sw.plank)]
    d_w = t(sw.weight)
    flux = torch.empty((W, 5, eng.nlev), dtype=torch.float64, device=dev)
    status = torch.empty(W, dtype=torch.int32, device=dev)
    acc = torch.zeros((5, eng.nlev), dtype=torch.float64, device=dev)
    tstream = torch.cuda.Stream(dev)                   # an explicit stream: copies, kernels and sums of a step in order
    torch.cuda.synchronize()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    import ctypes as C
    L = eng._L

    def step():
        eng.solve_device(*d_in, out=(flux, None, status), stream=stream)
        acc.zero_()
        rc = L.sbd_engine_accumulate_device(eng._h, W, d_w.data_ptr(), flux.data_ptr(), None,
                                            acc.data_ptr(), None, C.c_void_p(stream))
        assert rc == 0, rc
        coll.reduce_sum_(acc)                               # the one RCCL collective of the path

    def barrier():
        coll.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # HIP events around every kernel family of every pass, recorded on the streams the kernels are launched on (the
    # engine's two pass streams) DURING the timed steps and read after them: what the roofline object is computed from.
    # The mode is switched on and the events are CREATED here, in one more untimed step, so that no hipEventCreate falls
    # inside the timed region
    L.sbd_engine_enable_timing(eng._h, 2)
    step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    inplace_ms = np.array([eng.last_ms(p) for p in range(5)])       # (the last timed step's launches)
    L.sbd_engine_enable_timing(eng._h, 0)
    elapsed_rank = elapsed
    elapsed = coll.all_max(elapsed)                        # the slowest rank's clock
    W_all = int(round(coll.all_sum(float(W))))
    # every rank's shard and clock, for the line: [first point, one past the last (strong: of the ONE sweep; weak: of the
    # rank's own sweep), solves, seconds of the timed steps]
    shards = coll.gather(rank, [p_lo, p_hi, W, elapsed_rank])
    bad = int((status != 0).sum().item())
    if not bool(torch.isfinite(flux).all().item()):
        sys.exit("bench.py: non-finite fluxes -- refusing to report a rate for wrong answers")

    # ---- per-kernel timing pass (HIP events on the launch stream), outside the timed region: ONE stream, a
    #      synchronisation per pass -- each kernel family alone on the chip.  (--headline-only leaves it out: every launch
    #      of the command then ran in the timed configuration, which is what a kernel trace of it should average) ----
    phase_ms = np.array(inplace_ms, dtype=float)
    fallback_layers = -1
    if not args.headline_only:
        eng.enable_timing(True)
        phase_ms = np.zeros(5)
        nrep = 3
        for _ in range(nrep):
            eng.solve_device(*d_in, out=(flux, None, status), stream=stream)
            phase_ms += [eng.last_ms(p) for p in range(5)]
        phase_ms /= nrep
        fallback_layers = int(eng.last_fallback_layers())
        eng.enable_timing(False)
    torch.cuda.synchronize()

    # ---- the same step through the HOST entry points (what the Fortran host calls), outside the timed region; never `value` ----
    hl = None if args.headline_only else host_entry_legs(args, sw, eng, d_in, d_w, acc, flux, status, stream, dev, local_rank,
                                                         rank, world, level_out, barrier, strong, coll)

    if rank == 0:
        nwl_total = args.nwl if strong else sw.nwl * world
        W_total = W_all                                     # solves of the whole node per step
        ms_per_step = 1e3 * elapsed / args.steps
        value = nwl_total * args.steps / elapsed
        names = ["setup_kernel", "layer_kernel", "band_kernel", "backsolve_kernel", "usrint+azimuth"]
        abytes = algorithmic_bytes_per_solve(sw.nlyr, sw.nstr, eng.nlev)
        nlaunch = npass_headline                            # (sbd_engine_pass_count: an even number of equal passes,
                                                            #  alternating between the engine's two workspaces / streams)
        pass_size = (W + nlaunch - 1) // nlaunch
        # the dominant kernel's launches as they ran in the timed region (two streams: a pass's band LU beside the other
        # pass's layer kernel); `kernel_ms` below is the serialized pass (each kernel family alone on the chip)
        roof = shape_roofline(names, inplace_ms if (inplace_ms > 0).all() else phase_ms, nlaunch, pass_size, W, sw.nstr, sw.nlyr, eng.nlev,
                              fused=True)
        roof["timed_with"] = "HIP events on the engine's pass streams during the last timed step (sbd_engine_enable_timing 2)"
        roof["note"] = ("latency/issue bound by construction (SURVEY 8d): ~5 KB of inputs per 2.5 MFLOP of pivoted fp64; the "
                        "binding figures are roofline_fp64 (algorithmic flops against the fp64 vector peak) and valu_issue "
                        "(executed VALU occupancy), beside it")
        flops = algorithmic_flops_per_solve(sw.nlyr, sw.nstr)
        tf = flops * W_total * args.steps / elapsed / 1e12
        out = {
            "metric": "spectral-points/sec (whole node) + flux RMSE vs CPU, 16-stream SW sweep",
            "value": value, "unit": "spectral-points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"SW 0.25-4.0um synthetic sweep, nstr={sw.nstr}, {sw.nlyr} layers, "
                                   + (f"ONE sweep of {args.nwl} spectral points cut between points over {world} GPU(s) "
                                      f"(sbd_shard_range; rank 0: {sw.nwl} points, {W} solves), {W_total} DISORT solves, "
                                      if strong else
                                      f"{sw.nwl} spectral points/GPU, {W} DISORT solves/GPU (avg nk {W / sw.nwl:.2f}), ")
                                   + "flux at TOA+surface, seed 12345; `value` = rate with the inputs RESIDENT in HBM when the timed "
                                     "region starts (north_star: optical depths precomputed into HBM arrays), `value_8d` beside it = "
                                     "SURVEY 8(d)'s engine phase incl. H2D of the compact inputs and D2H of the sums",
                       "nstr": sw.nstr, "nlyr": sw.nlyr, "nwl_per_gpu": sw.nwl, "solves_per_gpu": W,
                       "nwl_total": nwl_total, "solves_total": W_total,
                       "shards": [{"rank": r_, "points": [int(a_), int(b_)], "solves": int(w_), "timed_s": t_}
                                  for r_, (a_, b_, w_, t_) in enumerate(shards)],
                       "parallelism": f"spectral shard x{world}, 1 RCCL reduce of {5 * eng.nlev} doubles/step; per GPU {nlaunch} passes alternating on 2 streams",
                       "chunk": eng.chunk, "workspace_bytes": eng.workspace_bytes},
            "solves_per_s": W_total * args.steps / elapsed,
            # SURVEY 8(d)'s fp64 fraction: ALGORITHMIC flops per solve (its formula, algorithmic_flops_per_solve) x the
            # solves of the timed steps / their time, against the fp64 VECTOR peak of the GPUs used (no MFMA on this path)
            "roofline_fp64": {"bound": "fp64_vector", "achieved": tf, "peak": FP64_VEC_PEAK_TF * world, "unit": "TFLOP/s",
                              "frac": tf / (FP64_VEC_PEAK_TF * world), "algorithmic_flops_per_solve": flops,
                              "note": "whole timed step (all kernels, all ranks), SURVEY 8(d)'s flop count; executed "
                                      "instructions are in valu_issue"},
            **(hl or {}),
            "nonzero_status": bad, "fallback_layers": fallback_layers,
            "kernel_ms": {names[i]: float(phase_ms[i]) for i in range(5)},
            "kernel_ms_in_timed_step": {names[i]: float(inplace_ms[i]) for i in range(5)},
            "roofline": roof,
        }
        out["valu_issue"] = valu_issue(W, elapsed / args.steps, sw.nstr, sw.nlyr)
        flux_h = flux.cpu().numpy()
        if share:
            out["rehearsal"] = True
            out["rehearsal_note"] = (f"--share-device: {world} ranks on ONE GPU over gloo -- exercises the N > 1 bookkeeping of this "
                                     "line (shards, MAX-reduced clock, sum over ranks); NOT a multi-GPU measurement")
        if world == 1 and not args.no_side_lines:
            eng.close()
            del d_in, flux, status
            torch.cuda.empty_cache()
            out["latency_case"] = latency_case(local_rank)
            out["other_shapes"] = other_shapes(dev)
            out["e2e_input_to_stdout"] = e2e_input_to_stdout(local_rank)
        if not args.no_cpu_baseline:
            # rank 0's host cores, rank 0's shard (N > 1: the same bounded sample of the same workload -- a per-core
            # figure does not depend on how many GPUs the sweep is cut over; `flux_rmse_vs_cpu` then checks rank 0's items)
            out["cpu_baseline"], allc, idx, ref = cpu_baseline(sw, seconds_target=args.cpu_baseline_seconds)
            if world > 1:
                out["cpu_baseline"]["sample"] = str(out["cpu_baseline"].get("sample", "")) + f" (rank 0's shard of {world})"
            if allc is not None:
                out["cpu_baseline_allcore"] = allc
            if ref is not None:
                # the other half of the metric: GPU vs CPU on the sampled solves, TOA (level 0) and
                # surface (level 1) -- per solve (fbeam = 1: fluxes per unit incident beam) and for
                # the spectrally weighted sums of the sample (stdout1's TOPDN..BOTDIR, drt.f:1047-1054)
                ns = ref.shape[0]
                g = flux_h[idx][:, :3, :]   # [ns][rfldir, rfldn, flup][top, bot]
                six = lambda a: np.stack([a[:, 1] + a[:, 0], a[:, 2], a[:, 0]], 1).reshape(ns, 6)   # dn, up, dir at top|bot
                sg, sr = six(g), six(ref)
                wgt = np.asarray(sw.weight[idx], dtype=np.float64)[:, None]
                out["flux_rmse_vs_cpu"] = {
                    "per_solve_rmse": float(np.sqrt(np.mean((sg - sr) ** 2))),
                    "per_solve_max_abs": float(np.abs(sg - sr).max()),
                    "per_solve_max_rel_to_max": float(np.abs(sg - sr).max() / np.abs(sr).max()),
                    "integrated_max_abs": float(np.abs(((sg - sr) * wgt).sum(0)).max()),
                    "integrated_ref_max": float(np.abs((sr * wgt).sum(0)).max()),
                    "quantities": "TOPDN,BOTDN,TOPUP,BOTUP,TOPDIR,BOTDIR", "solves": int(ns),
                    "thermal_solves": int(np.count_nonzero(np.asarray(sw.plank)[idx])),
                    "units": "fluxes per unit FBEAM (synthetic sweep has FBEAM = 1; thermal items in W/m2 per band); "
                             "north_star gate 1e-4 W/m2 on integrated fluxes"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
