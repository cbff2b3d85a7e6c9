"""Several engines behind one handle (sbd_fleet_*, include/sbdart_amd.h) on the GPU box.

The box has one GPU, so the fleet is built over the device list [0, 0]: two engines, two
workspaces, two streams, the batch cut by sbd_shard_range, the weighted sums combined at the end
(host-side here: RCCL does not take the same device twice; with distinct devices the same call
reduces over xGMI).  Per-item outputs must equal the single engine's bit for bit, the reduced
sums must equal stdout1's accumulation of the single engine's outputs."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _batch(recs):
    return (np.stack([r.dtauc for r in recs]), np.stack([r.ssalb for r in recs]), np.stack([r.pmom for r in recs]),
            [r.wvnmlo for r in recs], [r.wvnmhi for r in recs], [r.fbeam for r in recs],
            [r.albedo for r in recs], [r.plank for r in recs])


@pytest.mark.parametrize("devices,name", [([0, 0], "cfgB_sw_nstr16"), ([0, 0, 0], "cfg3_lw_nstr16_cloud"), ([0], "cfgA_sw_nstr4"),
                                          (None, "sbchk5")])
def test_fleet_matches_single_engine(devices, name):
    from sbdart_amd.engine import DisortFleet, engine_for_record
    from sbdart_amd.records import read_records
    recs = read_records(os.path.join(GOLDEN, name + ".sbdrec"))
    r0 = recs[0]
    recs = [r for r in recs if np.array_equal(r.temper, r0.temper) and r.umu0 == r0.umu0 and r.nstr == r0.nstr][:23]
    args = _batch(recs)
    w = np.array([r.wt * r.ff for r in recs])
    lev = None if not r0.onlyfl else [0, r0.nlyr]
    with engine_for_record(r0, level_out=lev) as one:
        f1, u1, s1 = one.solve(*args)
    kw = dict(nlyr=r0.nlyr, nstr=r0.nstr, nmom=r0.nmom, temper=r0.temper, umu0=r0.umu0, phi0=r0.phi0, onlyfl=r0.onlyfl,
              usrang=r0.usrang, umu=r0.umu, phi=r0.phi, btemp=r0.btemp, ttemp=r0.ttemp, temis=r0.temis, fisot=r0.fisot,
              level_out=lev, allow_retry_nstr=True)
    with DisortFleet(devices=devices, **kw) as fl:
        assert fl.size == (len(devices) if devices else 1)      # one GPU visible on the box
        parts = [fl.shard_range(len(recs), r) for r in range(fl.size)]
        assert parts[0][0] == 0 and parts[-1][1] == len(recs)
        f2, u2, s2, acc_f, acc_u = fl.solve(*args, weight=w)
        f3, u3, s3 = fl.solve(*args)                            # no sums asked for
        _, _, s4, acc_f4, _ = fl.solve(*args, weight=w, items=False)   # sums only
    assert np.array_equal(f1, f2) and np.array_equal(s1, s2) and np.array_equal(f1, f3)
    if u1 is not None:
        assert np.array_equal(u1, u2)
        assert np.allclose(acc_u, np.einsum("i,ipln->pln", w, u1), rtol=1e-12, atol=1e-300)
    assert np.allclose(acc_f, np.einsum("i,icl->cl", w, f1), rtol=1e-12, atol=1e-300)
    assert np.array_equal(acc_f, acc_f4) and np.array_equal(s1, s4)


def test_host_entry_point_with_growing_batches():
    """The host-pointer solve keeps a device staging area and a pinned landing buffer for the outputs, both
    grown on demand: a small batch, a larger one, the small one again -- each must equal the device-pointer
    results of the same items (several passes in the large one: the copy stream and its events)."""
    import torch
    from sbdart_amd.engine import DisortEngine
    from sbdart_amd.workload import sw_sweep
    sw = sw_sweep(nwl=1536, nstr=8, nlyr=20, seed=99, shard=0)
    ins = (sw.dtauc, sw.ssalb, sw.pmom, sw.wvnmlo, sw.wvnmhi, sw.fbeam, sw.albedo, sw.plank)
    cut = lambda n: tuple(a[:n] for a in ins)
    with DisortEngine(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp,
                      ttemp=sw.ttemp, temis=sw.temis, onlyfl=True, level_out=[0, sw.nlyr], device=0, max_batch=512) as eng:
        f_small, _, s_small = eng.solve(*cut(37))
        f_big, _, s_big = eng.solve(*ins)                      # several passes of <= 512 items
        f_again, _, s_again = eng.solve(*cut(37))
    assert sw.nwork > 2048
    assert np.isfinite(f_big).all() and (s_big == 0).all()
    assert np.array_equal(f_small, f_big[:37]) and np.array_equal(f_again, f_small)
    assert np.array_equal(s_small, s_big[:37]) and np.array_equal(s_again, s_small)


def test_fleet_feeds_its_devices_from_one_thread_each():
    """A pageable batch (numpy arrays: like a Fortran ALLOCATE) on the device list [0, 0, 0]: the three shards are
    enqueued from a host thread each and the large input arrays are page-locked for the call, so no device waits
    for another's staging.  The enqueue intervals (host clock, sbd_fleet_last_enqueue) must overlap -- with one
    enqueueing thread they would follow each other -- and the answers stay bitwise those of a single engine."""
    from sbdart_amd.engine import DisortEngine, DisortFleet
    from sbdart_amd.workload import sw_sweep
    sw = sw_sweep(nwl=9000, nstr=16, nlyr=33, seed=31)         # ~24 000 solves, 0.12 GB of pageable inputs
    ins = (sw.dtauc, sw.ssalb, sw.pmom, sw.wvnmlo, sw.wvnmhi, sw.fbeam, sw.albedo, sw.plank)
    kw = dict(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp,
              ttemp=sw.ttemp, temis=sw.temis, onlyfl=True, level_out=[0, sw.nlyr])
    with DisortEngine(device=0, **kw) as one:
        f1, _, s1 = one.solve(*ins)
    with DisortFleet(devices=[0, 0, 0], **kw) as fl:
        f2, _, s2, acc, _ = fl.solve(*ins, weight=sw.weight)
        spans, npinned = fl.last_enqueue()
        f3, _, s3, acc3, _ = fl.solve(*ins, weight=sw.weight)   # (registered and released again: same answers)
    assert np.array_equal(f1, f2) and np.array_equal(s1, s2) and np.array_equal(f2, f3) and np.array_equal(acc, acc3)
    assert npinned == 3, npinned
    assert len(spans) == 3 and all(e > b for b, e in spans)
    latest_begin, earliest_end = max(b for b, _ in spans), min(e for _, e in spans)
    assert latest_begin < earliest_end, spans                   # every device was being fed at the same time
    assert np.allclose(acc, np.einsum("i,icl->cl", sw.weight, f1), rtol=1e-12, atol=1e-300)


def test_rccl_reduce_path_on_one_device():
    """SBD_FLEET_RCCL=1: a fleet of ONE device takes the collective path -- librccl dlopen()ed, ncclCommInitAll
    over [0] (nranks = 1), the grouped ncclReduce(sum, double) of the accumulator block into d_red, its D2H, the
    communicator's teardown -- which a one-GPU box otherwise never executes.  Sums and items must equal the
    host-side-sum fleet's bit for bit."""
    import json, subprocess, sys
    from conftest import ROOT
    code = ("import numpy as np,json;from sbdart_amd.engine import DisortFleet;from sbdart_amd.workload import sw_sweep;"
            "sw=sw_sweep(nwl=600,nstr=16,nlyr=33,seed=5);"
            "kw=dict(nlyr=sw.nlyr,nstr=sw.nstr,nmom=sw.nmom,temper=sw.temper,umu0=sw.umu0,btemp=sw.btemp,ttemp=sw.ttemp,"
            "temis=sw.temis,onlyfl=True,level_out=[0,sw.nlyr]);"
            "fl=DisortFleet(devices=[0],**kw);"
            "f,_,s,acc,_=fl.solve(sw.dtauc,sw.ssalb,sw.pmom,sw.wvnmlo,sw.wvnmhi,sw.fbeam,sw.albedo,sw.plank,weight=sw.weight);"
            "f2,_,s2,acc2,_=fl.solve(sw.dtauc,sw.ssalb,sw.pmom,sw.wvnmlo,sw.wvnmhi,sw.fbeam,sw.albedo,sw.plank,weight=sw.weight);"
            "r=fl.uses_rccl;fl.close();"
            "print(json.dumps(dict(rccl=bool(r),acc=acc.tolist(),acc2=acc2.tolist(),fsum=float(np.abs(f).sum()),bad=int((s!=0).sum()))))")
    res = {}
    for flag in ("0", "1"):
        p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, text=True, capture_output=True,
                           env=dict(os.environ, SBD_FLEET_RCCL=flag, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"rccl"')]      # (RCCL may print banners of its own)
        assert p.returncode == 0 and lines, (flag, p.returncode, p.stdout[-2000:], p.stderr[-2000:])
        res[flag] = json.loads(lines[-1])
    assert res["0"]["rccl"] is False and res["1"]["rccl"] is True
    assert res["1"]["bad"] == 0
    assert res["1"]["acc"] == res["0"]["acc"] and res["1"]["acc2"] == res["1"]["acc"] and res["1"]["fsum"] == res["0"]["fsum"]


def test_moments_shared_by_the_k_terms_of_a_point():
    """sbd_batch_in::pmom_row: one block of phase-function moments per SPECTRAL POINT instead of per work item (the
    reference computes them once per wavelength, drt.f:476-533).  Same answers, bit for bit, as the expanded batch --
    through the host entry point (several passes: every pass copies the range of blocks its items point at), the
    device entry point, a three-engine fleet, and with unsorted rows (all blocks copied up front)."""
    import torch
    from sbdart_amd.engine import DisortEngine, DisortFleet
    from sbdart_amd.workload import sw_sweep
    sw = sw_sweep(nwl=700, nstr=8, nlyr=20, seed=17)
    rows = sw.wl_of.astype(np.int32)                            # item -> spectral point
    pm_pt = np.stack([sw.pmom[np.nonzero(rows == k)[0][0]] for k in range(sw.nwl)])      # the point's first item's moments
    pm_full = pm_pt[rows]
    ins = lambda pm: (sw.dtauc, sw.ssalb, pm, sw.wvnmlo, sw.wvnmhi, sw.fbeam, sw.albedo, sw.plank)
    kw = dict(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp, ttemp=sw.ttemp,
              temis=sw.temis, onlyfl=True, level_out=[0, sw.nlyr])
    with DisortEngine(device=0, max_batch=256, **kw) as eng:
        f0, _, s0 = eng.solve(*ins(pm_full))
        f1, _, s1 = eng.solve(*ins(pm_pt), pmom_row=rows)
        perm = np.random.default_rng(1).permutation(sw.nwork)
        f2, _, s2 = eng.solve(sw.dtauc[perm], sw.ssalb[perm], pm_pt, sw.wvnmlo[perm], sw.wvnmhi[perm], sw.fbeam[perm],
                              sw.albedo[perm], sw.plank[perm], pmom_row=rows[perm])
        dev = torch.device("cuda:0")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        f3, _, s3 = eng.solve(t(sw.dtauc), t(sw.ssalb), t(pm_pt), t(sw.wvnmlo), t(sw.wvnmhi), t(sw.fbeam), t(sw.albedo),
                              t(sw.plank), pmom_row=t(rows))
        torch.cuda.synchronize()
    assert (s0 == 0).all() and sw.nwork > 4 * 256
    assert np.array_equal(f0, f1) and np.array_equal(s0, s1)
    assert np.array_equal(f0[perm], f2) and np.array_equal(f0, f3.cpu().numpy())
    with DisortFleet(devices=[0, 0, 0], **kw) as fl:
        f4, _, s4, acc, _ = fl.solve(*ins(pm_pt), weight=sw.weight, pmom_row=rows)
    assert np.array_equal(f0, f4)


def test_growing_passes_of_the_host_entry_point():
    """With the moments shared per spectral point and 32 768 items or more, the host entry point starts with a pass of
    8 192 items and grows the passes by 1.3 x (the first inputs cross PCIe sooner; sbd_engine.hip, solve_device_impl).
    An item's result does not depend on the pass it rides in: bit for bit the answers of the device entry point (equal
    passes) and of the host entry point with per-item moments (equal passes)."""
    import torch
    from sbdart_amd.engine import DisortEngine
    from sbdart_amd.workload import sw_sweep
    sw = sw_sweep(nwl=13000, nstr=8, nlyr=12, seed=23)
    assert sw.nwork >= 32768
    rows = sw.wl_of.astype(np.int32)
    first = np.concatenate([[0], np.nonzero(np.diff(rows))[0] + 1])
    pm_pt = np.ascontiguousarray(sw.pmom[first])
    pm_full = pm_pt[rows]
    kw = dict(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp, ttemp=sw.ttemp,
              temis=sw.temis, onlyfl=True, level_out=[0, sw.nlyr])
    with DisortEngine(device=0, **kw) as eng:
        assert eng.chunk < sw.nwork or sw.nwork >= 32768
        f_ramp, _, s_ramp = eng.solve(sw.dtauc, sw.ssalb, pm_pt, sw.wvnmlo, sw.wvnmhi, sw.fbeam, sw.albedo, sw.plank, pmom_row=rows)
        f_host, _, s_host = eng.solve(sw.dtauc, sw.ssalb, pm_full, sw.wvnmlo, sw.wvnmhi, sw.fbeam, sw.albedo, sw.plank)
        dev = torch.device("cuda:0")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        f_dev, _, s_dev = eng.solve(t(sw.dtauc), t(sw.ssalb), t(pm_pt), t(sw.wvnmlo), t(sw.wvnmhi), t(sw.fbeam), t(sw.albedo),
                                    t(sw.plank), pmom_row=t(rows))
        torch.cuda.synchronize()
    assert (s_ramp == 0).all()
    assert np.array_equal(f_ramp, f_host) and np.array_equal(s_ramp, s_host)
    assert np.array_equal(f_ramp, f_dev.cpu().numpy()) and np.array_equal(s_ramp, s_dev.cpu().numpy())


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_compact_form_over_several_devices_cuts_between_points(devices):
    """sbd_fleet_solve_mix_host on a fleet of one, two and three engines (the box's one GPU listed again): the batch is cut
    by sbd_shard_range_points, every shard stages only the point blocks its items refer to (block indices counted from
    its first), and the per-item outputs and status words equal the one-engine call bit for bit; the sums agree to the
    association of their shard-wise addition."""
    from sbdart_amd.engine import DisortFleet
    from sbdart_amd.shard import shard_range_points
    from sbdart_amd.workload import sw_sweep_mix
    m = sw_sweep_mix(nwl=900, nstr=16, seed=99)
    kw = dict(nlyr=m.nlyr, nstr=m.nstr, nmom=m.nmom, temper=m.temper, umu0=m.umu0, btemp=m.btemp, ttemp=m.ttemp,
              temis=m.temis, onlyfl=True, level_out=[0, m.nlyr])
    with DisortFleet(devices=[0], **kw) as one:
        f1, _, s1, a1, _ = one.solve_mix(*m.mix_args(), weight=m.weight)
    with DisortFleet(devices=devices, **kw) as fl:
        f2, _, s2, a2, _ = fl.solve_mix(*m.mix_args(), weight=m.weight)
        # a part of a run: items of the points 300.. only, block indices global (the host's beam / no-beam parts)
        lo = int(np.searchsorted(m.point_of, 300))
        f3, _, s3 = fl.solve_mix(m.point_of[lo:], m.dtaug[lo:], m.lay, m.family, m.wvnmlo, m.wvnmhi, m.fbeam, m.albedo, m.plank)
    assert np.array_equal(f1, f2) and np.array_equal(s1, s2) and (s1 == 0).all()
    assert np.array_equal(f1[lo:], f3) and np.array_equal(s1[lo:], s3)
    assert np.allclose(a1, a2, rtol=1e-13, atol=0)
    cuts = [shard_range_points(m.point_of, r, len(devices)) for r in range(len(devices))]
    assert all(a == 0 or m.point_of[a] != m.point_of[a - 1] for a, _ in cuts)
