"""The N>1 path on CPU: two gloo ranks cut a small sweep with the product's own shard rule (the
C ABI's sbd_shard_range, the function the fleet and the Fortran host shard with -- host code,
callable without a GPU), each integrates its shard (the C oracle standing in for the GPU
solve: without a GPU this test is about the sharding and the single reduce, not the kernels)
and the reduced accumulators equal the single-process sums.  The same split with real engines
on a GPU: tests/test_gpu_fleet.py."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ORACLE_DIR, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _integrate(sw, lo, hi):
    import pyoracle
    from sbdart_amd.workload import sweep_to_records
    idx = np.nonzero((sw.wl_of >= lo) & (sw.wl_of < hi))[0]
    acc = np.zeros((5, 2))
    for i, rec in zip(idx, sweep_to_records(sw, idx)):
        o = pyoracle.disort(rec)
        for c, f in enumerate(("rfldir", "rfldn", "flup", "dfdt", "uavg")):
            acc[c] += sw.weight[i] * o[f][[0, -1]]
    return acc


def _integrate_gpu(sw, lo, hi):
    """The same shard through the HIP engine (device-side weighted sums, sbd_engine_accumulate_*)."""
    from sbdart_amd.engine import DisortEngine
    idx = np.nonzero((sw.wl_of >= lo) & (sw.wl_of < hi))[0]
    with DisortEngine(nlyr=sw.nlyr, nstr=sw.nstr, nmom=sw.nmom, temper=sw.temper, umu0=sw.umu0, btemp=sw.btemp,
                      ttemp=sw.ttemp, temis=sw.temis, onlyfl=True, level_out=[0, sw.nlyr], device=0) as eng:
        flux, _, st = eng.solve(sw.dtauc[idx], sw.ssalb[idx], sw.pmom[idx], sw.wvnmlo[idx], sw.wvnmhi[idx],
                                sw.fbeam[idx], sw.albedo[idx], sw.plank[idx])
        assert (st == 0).all()
        acc, _ = eng.accumulate(sw.weight[idx], flux)
    return acc


def _worker(rank, world, port, q, engine=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, ORACLE_DIR)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C
    from sbdart_amd import _lib
    from sbdart_amd.shard import reduce_accumulators
    from sbdart_amd.workload import sw_sweep
    sw = sw_sweep(nwl=12, nstr=8, nlyr=5, seed=99)
    clo, chi = C.c_int32(), C.c_int32()
    _lib.load().sbd_shard_range(sw.nwl, world, rank, C.byref(clo), C.byref(chi))
    lo, hi = clo.value, chi.value
    if engine:
        torch.cuda.init()                                   # PyTorch's HIP runtime before the engine's (conftest.py)
    acc = torch.from_numpy(np.ascontiguousarray(_integrate_gpu(sw, lo, hi) if engine else _integrate(sw, lo, hi)))
    reduce_accumulators(acc, dst=0)
    if rank == 0:
        q.put(acc.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_rank_spectral_shard_and_reduce_with_engines():
    """The same two-rank run with the HIP engine solving each rank's shard (both ranks on the box's one GPU; the
    reduce stays on gloo -- RCCL refuses two ranks on one device): sharding rule + engine + reduce together,
    against the C oracle's whole-sweep integral."""
    _two_ranks(engine=True, rtol=5e-6)     # (the engine-vs-reference gate of tests/test_gpu_parity.py)


def test_two_rank_spectral_shard_and_reduce():
    _two_ranks(engine=False, rtol=1e-13)


def _two_ranks(engine, rtol):
    sys.path.insert(0, ORACLE_DIR)
    from sbdart_amd.workload import sw_sweep
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, engine)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sw = sw_sweep(nwl=12, nstr=8, nlyr=5, seed=99)
    whole = _integrate(sw, 0, sw.nwl)
    assert np.allclose(got, whole, rtol=rtol, atol=rtol * 1e-3 * np.abs(whole).max())
