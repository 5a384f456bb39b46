"""Multi-GPU decomposition: the env batch is sharded contiguously across ranks (one process per GPU); buildings of
an env never leave their GPU, so every district reduction is intra-workgroup and the step path needs no collective
(SURVEY.md 8e).  The only cross-rank traffic is benchmark / logging scalars.

Besides the shard arithmetic this module holds the plumbing `bench.py` uses to be started by a plain
``python bench.py --gpus N`` (no ``torch.distributed.run``): `launch_ranks` starts one process per rank with the
usual RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment, `init_control_plane` brings up the process group those
ranks use for the barrier and the MAX-over-ranks of the timing -- RCCL (``nccl``) with one rank per GPU, ``gloo``
when ranks share a device (RCCL refuses two ranks on one GPU; only the 1-GPU test hook does that) or on CPU."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List, Mapping, Optional, Sequence, Tuple


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


def gather_seconds(seconds: float, dist=None, device=None) -> List[float]:
    """The same measurement of every rank, in rank order (bench.py's per-rank `ms_per_step`)."""
    if dist is None or not dist.is_initialized():
        return [float(seconds)]
    import torch
    mine = torch.tensor([seconds], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t[0]) for t in out]


def free_port() -> int:
    """A TCP port nobody listens on right now (127.0.0.1), for the rendezvous of self-spawned ranks."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return int(s.getsockname()[1])


def rank_environment(rank: int, world: int, port: int, base: Optional[Mapping[str, str]] = None) -> dict:
    """Environment of rank `rank` of a one-node job: what ``torch.distributed.run --nnodes=1`` would set."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL across processes needs it on this driver
    return env


def launch_ranks(argv: Sequence[str], world: int, timeout: Optional[float] = None,
                 extra_env: Optional[Mapping[str, str]] = None) -> Tuple[int, str]:
    """Start `world` processes ``argv`` (one per rank; rank r gets RANK = LOCAL_RANK = r) and wait for them.

    Rank 0's stdout is captured and returned; the other ranks' stdout goes to this process's stderr, every rank's stderr is
    inherited.  Returns ``(exit code, rank 0's stdout)``: the first non-zero exit code of any rank (the remaining ranks are
    then terminated -- by the PIDs started here, never by pattern), 124 after `timeout` seconds, else 0."""
    if world < 1:
        raise ValueError('world must be >= 1')
    port = free_port()
    procs = []
    for r in range(world):
        env = rank_environment(r, world, port)
        env.update(extra_env or {})
        # (ranks > 0: stdout onto this process's stderr -- file descriptor 2, whatever object sys.stderr currently is)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=subprocess.PIPE if r == 0 else 2, stderr=None, text=True))
    t0, rc = time.monotonic(), 0
    out0 = ''
    try:
        # rank 0's pipe is drained by communicate(); a failing peer would leave rank 0 hanging in a collective, so the
        # peers are polled while we wait
        drained = False
        while True:
            if not drained:
                try:
                    out0, _ = procs[0].communicate(timeout=0.5)
                    drained = True
                except subprocess.TimeoutExpired:
                    pass
            else:
                time.sleep(0.1)
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                rc = 124
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        if not drained:
            try:
                out0, _ = procs[0].communicate(timeout=10)      # (keeps what the timed-out calls above had already read)
            except Exception:
                pass
    return rc, out0


class ControlPlane:
    """`torch.distributed` (every attribute is forwarded) + what `init_control_plane` settled on: `control_backend` ('nccl' = RCCL, or 'gloo')
    and `control_fallback` (why RCCL was given up, else None)."""

    def __init__(self, dist, backend: str, fallback: Optional[str] = None):
        self._dist, self.control_backend, self.control_fallback = dist, backend, fallback

    def __getattr__(self, name):
        return getattr(self._dist, name)


def init_control_plane(rank: int, world: int, device=None, backend: Optional[str] = None) -> ControlPlane:
    """Process group for the benchmark's barrier + timing reductions.  ``backend``: 'nccl' (RCCL over xGMI, one rank per GPU),
    'gloo' (ranks sharing a device, or CPU); default: nccl when `device` is a GPU.  Returns a `ControlPlane` (``torch.distributed`` + the
    backend that ended up carrying the barrier).

    RCCL prints a version banner on STDOUT when the communicator comes up; stdout is parked on stderr meanwhile so that it
    carries nothing but the caller's own output."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if world == 1:
        os.environ.setdefault('MASTER_PORT', str(free_port()))          # (a lone rank has nobody to agree with; launchers set it for the others)
    if backend is None:
        backend = 'nccl' if (device is not None and torch.device(device).type == 'cuda') else 'gloo'
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    fallback = None
    try:
        if backend == 'nccl':
            try:
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(device))
                dist.barrier()
                torch.cuda.synchronize()
            except Exception as exc:                  # noqa: BLE001 -- whatever RCCL raises (it fails symmetrically: every rank lands here)
                # The step path has no collective: the group only carries the benchmark's barrier and a MAX over ranks.  If RCCL cannot
                # come up on this node (IPC mode, fabric, driver), those two travel over gloo instead and the line says so -- a scaling
                # run must not be lost to the control plane.  CL_BENCH_STRICT_RCCL=1 keeps the failure fatal.
                if os.environ.get('CL_BENCH_STRICT_RCCL') == '1':
                    raise
                print(f'[rank {rank}] RCCL control plane failed ({type(exc).__name__}: {exc}); falling back to gloo', file=sys.stderr, flush=True)
                try:
                    dist.destroy_process_group()
                except Exception:                     # noqa: BLE001
                    pass
                # a fresh rendezvous store on the next port, hosted by rank 0 itself: under torch.distributed.run the env:// store lives in
                # the launcher's agent (TORCHELASTIC_USE_AGENT_STORE) and nobody would serve another port
                from datetime import timedelta
                port = int(os.environ.get('MASTER_PORT', '29500')) + 1
                store = dist.TCPStore(os.environ['MASTER_ADDR'], port, world, rank == 0, timeout=timedelta(seconds=180))
                backend = 'gloo'
                dist.init_process_group('gloo', store=store, rank=rank, world_size=world)
                dist.barrier()
                fallback = f'{type(exc).__name__}: {exc}'[:300]
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            dist.barrier()
    finally:
        sys.stdout.flush()
        try:
            # RCCL's banner goes through C stdio, which buffers when stdout is a pipe or a file: without this flush it would come out at
            # process exit, on the restored descriptor -- after the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                             # noqa: BLE001
            pass
        os.dup2(saved, 1)
        os.close(saved)
    return ControlPlane(dist, backend, fallback)
