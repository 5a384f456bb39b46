"""Multi-GPU path = env-batch sharding with no data-path collective.  Covered on CPU with world_size-2 gloo:
the shard arithmetic, the barrier / MAX-over-ranks timing reduction bench.py uses, and that shards are disjoint
and cover the batch."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from citylearn_amd.parallel import shard_envs, reduce_max_seconds

ROOT = Path(__file__).resolve().parent.parent


def test_shard_envs_partition():
    for total, world in ((65536, 8), (1000, 3), (262144, 8), (64, 8), (12, 5)):
        shards = [shard_envs(total, r, world) for r in range(world)]
        assert shards[0][0] == 0 and shards[-1][1] == total
        for (a0, a1), (b0, b1) in zip(shards, shards[1:]):
            assert a1 == b0 and a1 >= a0
        assert all((s1 - s0) % 4 == 0 for s0, s1 in shards[:-1])            # kernel needs multiples of 4 per shard


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_envs(1000, rank, world)
    mine = torch.zeros(1000)
    mine[lo:hi] = 1
    dist.all_reduce(mine)                                                   # test-only check: shards tile the batch
    wall = reduce_max_seconds(0.5 + rank, dist)
    q.put((rank, lo, hi, float(mine.min()), float(mine.max()), wall))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_timing():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:3] == (0, 500) and res[1][1:3] == (500, 1000)
    assert all(r[3] == 1.0 and r[4] == 1.0 for r in res)                    # disjoint and complete
    assert all(abs(r[5] - 1.5) < 1e-9 for r in res)                         # MAX over ranks
