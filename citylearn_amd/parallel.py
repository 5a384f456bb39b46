"""Multi-GPU decomposition: the env batch is sharded contiguously across ranks (one process per GPU); buildings of
an env never leave their GPU, so every district reduction is intra-workgroup and the step path needs no collective
(SURVEY.md 8e).  The only cross-rank traffic is benchmark / logging scalars."""
from __future__ import annotations

from typing import Tuple


def shard_envs(total_envs: int, rank: int, world: int, align: int = 4) -> Tuple[int, int]:
    """Contiguous [start, end) env range of `rank`; all shards but the last are multiples of `align` envs."""
    per = -(-total_envs // world)
    per = -(-per // align) * align
    start = min(rank * per, total_envs)
    end = min(start + per, total_envs) if rank < world - 1 else total_envs
    return start, max(start, end)


def reduce_max_seconds(seconds: float, dist=None, device=None) -> float:
    """MAX over ranks of a wall-clock measurement (what bench.py reports)."""
    if dist is None or not dist.is_initialized():
        return float(seconds)
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
