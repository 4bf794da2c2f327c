"""One process per GPU, started by the program itself.

The reference launches its drivers with ``torch.multiprocessing.spawn`` over a TCP rendezvous
(slowfast/utils/misc.py:402-430 -> slowfast/utils/multiprocessing.py:8-62: ``launch_job`` spawns NUM_GPUS
children, each calls ``init_process_group(init_method="tcp://localhost:9999", world_size, rank)`` and then the
driver function).  Same process model here, with two differences: the parent picks a free port instead of a fixed
one, and it must not have touched the GPU before the children start (a spawned child is a fresh interpreter;
the parent only waits).  Under ``torch.distributed.run`` the ranks already exist and nothing is spawned.

    run(worker, n_ranks, args)   in the parent: returns after all ranks have finished (or raises what a rank raised)
    rank_env()                   in a rank: (rank, local_rank, world) from the environment
    init_process_group(...)      in a rank: RCCL ("nccl") or gloo over 127.0.0.1, checks the world size
"""
from __future__ import annotations

import os
import socket
from typing import Callable, Optional, Sequence, Tuple

_ENV_KEYS = ("RANK", "LOCAL_RANK", "WORLD_SIZE")


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def under_launcher() -> bool:
    """True when this process is already one rank of a job (torch.distributed.run or our own spawn)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def rank_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world); (0, 0, 1) for a plain single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if not 0 <= rank < world:
        raise SystemExit(f"RANK={rank} outside WORLD_SIZE={world}")
    return rank, local, world


def check_world(requested: int) -> Tuple[int, int, int]:
    """rank_env(), refusing a job whose size is not the one asked for on the command line."""
    rank, local, world = rank_env()
    if world != int(requested):
        raise SystemExit(f"launched with WORLD_SIZE={world} but {requested} ranks were asked for "
                         "(--gpus / NUM_GPUS must equal the number of ranks)")
    return rank, local, world


def _rank_main(local_rank: int, worker: Callable, world: int, addr: str, port: int, args: Sequence) -> None:
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world),
                       "LOCAL_WORLD_SIZE": str(world), "MASTER_ADDR": addr, "MASTER_PORT": str(port)})
    worker(*args)


def run(worker: Callable, n_ranks: int, args: Sequence = (), port: Optional[int] = None) -> None:
    """Run ``worker(*args)`` as ``n_ranks`` ranks of one node.  n_ranks == 1 (or an existing launcher
    environment): called in this process.  Otherwise the ranks are spawned as fresh interpreters with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set; the parent issues no GPU call (importing
    torch and counting devices is not one) and joins them.  `worker` must be a module-level function."""
    n_ranks = int(n_ranks)
    if n_ranks < 1:
        raise SystemExit(f"{n_ranks} ranks asked for")
    if under_launcher() or n_ranks == 1:
        worker(*args)
        return
    import torch.multiprocessing as mp
    port = free_port() if port is None else int(port)
    mp.spawn(_rank_main, args=(worker, n_ranks, "127.0.0.1", port, tuple(args)), nprocs=n_ranks, join=True)


class _StdoutToStderr:
    """File descriptor 1 points at stderr inside the block: gloo announces its connections with printf on stdout,
    and the drivers' stdout is reserved for the ONE result line."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def init_process_group(backend: str, device=None, force: bool = False) -> None:
    """Join the job's process group (no-op for a single process).  backend "nccl" is RCCL on ROCm.
    force (or TOME_FORCE_PROCESS_GROUP=1): a single process forms a group of one as well -- the way to run RCCL's
    initialisation and the job's one collective on a box with a single GPU (tests/test_launch_gpu.py)."""
    import torch.distributed as dist
    _, _, world = rank_env()
    force = force or os.environ.get("TOME_FORCE_PROCESS_GROUP", "0") == "1"
    if (world == 1 and not force) or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if backend == "nccl" and device is not None:
        dist.init_process_group(backend="nccl", device_id=device)
    else:
        with _StdoutToStderr():
            dist.init_process_group(backend=backend)
            dist.barrier()  # gloo connects (and reports) on first use
    if dist.get_world_size() != world:
        raise SystemExit(f"process group has {dist.get_world_size()} ranks, environment says {world}")


def census(device) -> dict:
    """What the job actually consists of: ranks that answered (all-reduce of ones) and the device index every
    rank computes on (all-gather) -- reported in the result lines so a reader can see that N ranks on N
    devices produced them."""
    import torch
    import torch.distributed as dist
    idx = device.index if getattr(device, "index", None) is not None else -1
    if not (dist.is_available() and dist.is_initialized()):
        return {"ranks_seen": 1, "devices": [idx]}
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    mine = torch.tensor([idx], dtype=torch.int64, device=device)
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    return {"ranks_seen": int(one.item()), "devices": [int(t.item()) for t in every]}


def require_one_gpu_per_rank(backend: str, world: int) -> None:
    """RCCL wants one GPU per rank: several ranks stacked on one device hang or fail inside the library instead of saying
    so.  Refused here, before the process group exists (`--backend gloo` / TOME_DIST_BACKEND=gloo shares a GPU for a dry
    run of the N > 1 code path).  Counting devices does not initialise the GPU."""
    import torch
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if backend == "nccl" and world > 1 and local_world > torch.cuda.device_count():
        raise SystemExit(f"{local_world} ranks on this node but {torch.cuda.device_count()} GPU(s): RCCL needs one GPU "
                         "per rank (the gloo backend shares a GPU for a dry run)")


def check_census(seen: dict, backend: str, world: int) -> None:
    """After the job: every rank answered, and under RCCL the ranks of one node computed on DISTINCT devices."""
    if seen["ranks_seen"] != world:
        raise SystemExit(f"{seen['ranks_seen']} ranks answered, {world} expected")
    single_node = int(os.environ.get("LOCAL_WORLD_SIZE", world)) == world
    if backend == "nccl" and world > 1 and single_node and len(set(seen["devices"])) != world:
        raise SystemExit(f"RCCL job of {world} ranks ran on devices {seen['devices']}: not one GPU per rank")

