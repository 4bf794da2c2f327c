"""Batch-data-parallel evaluation glue: clips are independent, every rank runs a
full replica on its own clips and the only communication is ONE all-reduce of
``[top1_correct, top5_correct, clips]`` (reference pattern: tools/train_net.py:515-522 ->
slowfast/utils/distributed.py:47-63; top-k as slowfast/utils/metrics.py:9-41)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of `total` clips for `rank`; sizes differ by at most one."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def topk_counts(logits: torch.Tensor, labels: torch.Tensor, ks=(1, 5)) -> torch.Tensor:
    """int64 [len(ks)+1]: number of clips whose label is among the top-k logits, then the clip count."""
    kmax = max(ks)
    top = logits.float().topk(kmax, dim=1).indices
    hit = top == labels[:, None]
    out = [hit[:, :k].any(dim=1).sum() for k in ks]
    out.append(torch.tensor(logits.shape[0], device=logits.device))
    return torch.stack([o.to(torch.int64) for o in out])


def all_reduce_counts(counts: torch.Tensor) -> torch.Tensor:
    """The single collective of the eval path (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts
