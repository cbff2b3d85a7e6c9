"""One process per GPU: how `bench.py --gpus N` gets its N ranks.

Two ways in, one result:
  * the driver's way: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`
    -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are in the environment, `rendezvous` joins;
  * the bare way: `python bench.py --gpus N` -- no WORLD_SIZE: `relaunch_one_rank_per_gpu` re-runs the same
    command line under torch.distributed.run (127.0.0.1, a free port) and hands back its exit code.
Either way `rendezvous` REFUSES a world that is not the N that was asked for and counts the ranks through the
backend itself (an all-reduce of ones: RCCL on GPUs, gloo in the CPU test), so a line labelled `n_gpus: N` exists
only if N ranks met.  Host logic, no GPU needed to test it (tests/test_bench_launcher.py).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Tuple


def launched_world() -> Optional[int]:
    """WORLD_SIZE when this process was started by a launcher, else None."""
    w = os.environ.get("WORLD_SIZE")
    return int(w) if w is not None and "RANK" in os.environ else None


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(nproc: int, script: str, argv: List[str], port: Optional[int] = None) -> List[str]:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def relaunch_one_rank_per_gpu(nproc: int, script: str, argv: List[str]) -> int:
    """Run `script argv` as nproc ranks on this node; returns the launcher's exit code (non-zero if any rank failed)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(launcher_command(nproc, script, argv), env=env)


def rendezvous(ngpus: int, backend: str = "nccl") -> Tuple[int, int, int]:
    """Join the job the environment describes; returns (rank, local_rank, world).  Raises SystemExit unless the world
    is exactly `ngpus` ranks AND the backend's own sum over the ranks says so."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = launched_world() or 1
    if world != ngpus:
        sys.exit(f"bench.py: --gpus {ngpus} but the launcher started {world} rank(s): refusing to mislabel the line")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            if torch.cuda.device_count() <= local_rank:
                sys.exit(f"bench.py: rank {rank} has no GPU {local_rank} ({torch.cuda.device_count()} visible)")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            one = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
            one = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        counted = int(round(float(one.item())))
        if dist.get_world_size() != ngpus or counted != ngpus:
            sys.exit(f"bench.py: {backend} counted {counted} rank(s) in a world of {dist.get_world_size()}, "
                     f"--gpus says {ngpus}")
    return rank, local_rank, world
