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


def _oracle_worker(rank, world, port, q):
    """Each rank steps ITS env shard of one district with the CPU oracle (the GPU engine's stand-in on a box without GPUs: same
    table packer, same action layout, same per-env independence) and ships its planes back; rank ranges come from shard_envs."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT / 'tests'))
    import numpy as np
    from golden_util import golden
    from oracle.c_oracle import COracle, OO, OS
    g = golden('g2022_all')
    spec = g.spec()
    tab = spec.episode_tables(0)
    E_total, K = 40, 12
    lo, hi = shard_envs(E_total, rank, world)
    acts = np.random.RandomState(3).uniform(-1, 1, size=(K, 17, E_total)).astype(np.float32)       # same global actions on every rank
    ora = COracle(spec, tab, hi - lo, reward='MARL')
    for t in range(K):
        out, out_env = ora.step(np.ascontiguousarray(acts[t][:, lo:hi]), t)
    mine = torch.zeros((3, 17, E_total), dtype=torch.float64)
    mine[0, :, lo:hi] = torch.from_numpy(ora.state[:, :, OS['SOC']].T.copy())
    mine[1, :, lo:hi] = torch.from_numpy(out[:, :, OO['NET']].T.copy())
    mine[2, :, lo:hi] = torch.from_numpy(out[:, :, OO['REWARD']].T.copy())
    dist.all_reduce(mine)                                                  # test-only gather: shards are disjoint, so the sum tiles them
    if rank == 0:
        q.put(mine.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_real_shards_tile_the_unsharded_district():
    """The multi-GPU decomposition on real work: two ranks step two `shard_envs` shards of a 40-env 2022 district for 12 steps (MARL
    reward: the only coupling is INSIDE an env, across its buildings) and together reproduce, bit for bit, the unsharded run --
    no value ever crosses a shard boundary, which is why the step path needs no collective."""
    import numpy as np
    sys.path.insert(0, str(ROOT / 'tests'))
    from golden_util import golden
    from oracle.c_oracle import COracle, OO, OS
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_oracle_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tiled = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = golden('g2022_all')
    spec = g.spec()
    E_total, K = 40, 12
    acts = np.random.RandomState(3).uniform(-1, 1, size=(K, 17, E_total)).astype(np.float32)
    ora = COracle(spec, spec.episode_tables(0), E_total, reward='MARL')
    for t in range(K):
        out, _ = ora.step(acts[t], t)
    assert np.array_equal(tiled[0], ora.state[:, :, OS['SOC']].T)
    assert np.array_equal(tiled[1], out[:, :, OO['NET']].T) and np.array_equal(tiled[2], out[:, :, OO['REWARD']].T)


# ---------------------------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` without torch.distributed.run: the launcher, the rendezvous and the aggregation, on CPU
# (CL_BENCH_DRY_RUN skips the GPU work only; tests/test_gpu_bench.py runs the real thing with two ranks on one GPU)
def _bench(*argv, env=None, timeout=300):
    import subprocess
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / 'bench.py'), *argv], env=e, capture_output=True, text=True, timeout=timeout)


def _no_gpu_here():
    import torch
    return not torch.cuda.is_available()


_dry = pytest.mark.skipif(not _no_gpu_here(), reason='CL_BENCH_DRY_RUN is refused on a box with a GPU (tests/test_gpu_bench.py covers that)')


@_dry
def test_bench_spawns_its_own_ranks():
    """The driver's plain command: two ranks come up (gloo here), every rank is timed, the line carries the MAX over ranks, and
    stdout holds nothing but the one JSON line."""
    import json
    p = _bench('--gpus', '2', '--steps', '20', '--warmup', '5', env={'CL_BENCH_DRY_RUN': '1'})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['world_size_seen'] == 2 and out['steps'] == 20 and out['warmup'] == 5
    assert out['rank_ms_per_step'] == [1.0, 2.0] and out['ms_per_step'] == 2.0            # MAX over ranks
    assert out['scaling'] == 'weak' and out['unit'] == 'building-timesteps/s'
    one = json.loads(_bench('--steps', '20', '--warmup', '5', env={'CL_BENCH_DRY_RUN': '1'}).stdout)
    assert out['value'] == one['value']          # twice the units in twice the (synthetic) time


@_dry
def test_bench_eight_ranks_like_the_scaling_run():
    """The driver's N = 8 command shapes with the dry-run rank body: eight self-spawned ranks, and eight ranks under torch.distributed.run --
    one line on stdout, every rank's timing in rank order, MAX over ranks."""
    import json
    import subprocess
    from citylearn_amd.parallel import free_port
    p = _bench('--gpus', '8', '--steps', '20', '--warmup', '5', env={'CL_BENCH_DRY_RUN': '1'}, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(p.stdout.strip().splitlines()) == 1 and out['n_gpus'] == 8 and out['world_size_seen'] == 8
    assert out['rank_ms_per_step'] == pytest.approx([float(r + 1) for r in range(8)]) and out['ms_per_step'] == pytest.approx(8.0)
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    e['CL_BENCH_DRY_RUN'] = '1'
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
                        '--master-port', str(free_port()), str(ROOT / 'bench.py'), '--gpus', '8', '--steps', '20', '--warmup', '5'],
                       env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0])['rank_ms_per_step'] == pytest.approx([float(r + 1) for r in range(8)])


@_dry
def test_bench_under_torchrun_leaves_one_line_on_the_merged_stdout():
    """What the driver runs for N > 1: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`.  The launcher merges the
    ranks' stdout; only rank 0 may write there (the other ranks park descriptor 1 on stderr) -- here with the dry-run rank body."""
    import json
    import subprocess
    from citylearn_amd.parallel import free_port
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    e['CL_BENCH_DRY_RUN'] = '1'
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(free_port()), str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5'],
                       env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['world_size_seen'] == 2 and out['rank_ms_per_step'] == [1.0, 2.0]


def test_bench_under_an_external_launcher_is_one_rank():
    """With RANK / WORLD_SIZE in the environment (torch.distributed.run) bench.py must not spawn anything: WORLD_SIZE has to match --gpus."""
    p = _bench('--gpus', '2', '--steps', '20', '--warmup', '5', env={'CL_BENCH_DRY_RUN': '1', 'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '4',
                                                                      'MASTER_PORT': '1'})
    assert p.returncode != 0 and 'WORLD_SIZE=4' in p.stderr


def test_launcher_propagates_a_failing_rank_and_stops_the_others():
    import time
    from citylearn_amd.parallel import launch_ranks
    code = "import os, sys, time\nprint('hello from', os.environ['RANK'], flush=True)\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(120)"
    t0 = time.monotonic()
    rc, out0 = launch_ranks([sys.executable, '-c', code], 2)
    assert rc == 3 and time.monotonic() - t0 < 60 and 'hello from 0' in out0
    code = "import os\nassert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == os.environ['RANK'] and os.environ['MASTER_ADDR'] == '127.0.0.1'\nprint(os.environ['RANK'])"
    rc, out0 = launch_ranks([sys.executable, '-c', code], 3)
    assert rc == 0 and out0.strip() == '0'
    rc, _ = launch_ranks([sys.executable, '-c', 'import time; time.sleep(120)'], 2, timeout=2)
    assert rc == 124


def test_control_plane_falls_back_to_gloo_when_rccl_cannot_come_up():
    """`init_control_plane(.., 'nccl')` on a box where RCCL cannot build its communicator (here: no GPU at all; on a GPU node: IPC mode,
    fabric, two ranks on one device): every rank lands in the same `except`, the barrier and the MAX over ranks travel over gloo on a fresh
    rendezvous, and the caller can tell (`control_backend`, `control_fallback`).  CL_BENCH_STRICT_RCCL=1 keeps the failure fatal."""
    from citylearn_amd.parallel import launch_ranks
    code = ("import os, sys\n"
            f"sys.path.insert(0, {str(ROOT)!r})\n"
            "from citylearn_amd.parallel import init_control_plane, reduce_max_seconds\n"
            "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
            "d = init_control_plane(r, w, 'cuda:0', 'nccl')\n"
            "assert d.control_backend == 'gloo' and d.control_fallback, d.control_backend\n"
            "m = reduce_max_seconds(1.0 + r, d, 'cpu')\n"
            "d.barrier(); d.destroy_process_group()\n"
            "print('max', m, flush=True)\n")
    rc, out0 = launch_ranks([sys.executable, '-c', code], 2, timeout=240)
    assert rc == 0 and out0.strip().endswith('max 2.0'), out0
    rc, _ = launch_ranks([sys.executable, '-c', code], 2, timeout=240, extra_env={'CL_BENCH_STRICT_RCCL': '1'})
    assert rc not in (0, 124)
    # ... and under torch.distributed.run, whose env:// store lives in the launcher's agent: the fallback has to host its own store
    import subprocess
    from citylearn_amd.parallel import free_port
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(free_port()), '--no-python', sys.executable, '-c', code],
                       env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.count('max 2.0') == 2, (p.stdout[-500:], p.stderr[-1500:])


def test_bench_uses_only_counters_collected_on_the_kernel_it_launched(tmp_path, monkeypatch):
    """`bench._pmc_traffic`: a PMC summary under profiles/ feeds `roofline.traffic` only if its `_kernel.kernel` names the kernel the run
    launched (stale counters beside a newer kernel); `scripts/check_profiles.py` is the same check for a profile run."""
    import json
    import subprocess
    sys.path.insert(0, str(ROOT))
    import bench
    prof = tmp_path / 'profiles'
    prof.mkdir()
    summary = {'FETCH_SIZE': {'mean': 100.0}, 'WRITE_SIZE': {'mean': 50.0},
               '_kernel': {'kernel': 'void (anonymous namespace)::cl_step_envmajor_kernel<20>((anonymous namespace)::StepArgs)'}}
    # (file names carry the CURRENT round's prefix: since round 6 `_pmc_traffic` refuses summaries of earlier rounds -- VERDICT r05 item 10)
    cur = bench.ROUND_PREFIX
    (prof / f'{cur}a_streaming_pmc_summary.json').write_text(json.dumps(summary))
    (prof / 'r02_streaming_pmc_summary.json').write_text(json.dumps({**summary, '_kernel': {'kernel': 'cl_step_envmajor_kernel<20, true>'}}))   # stale round, right kernel: ignored
    monkeypatch.setattr(bench, 'ROOT', tmp_path)
    assert bench._pmc_traffic('r*_streaming_pmc_summary.json', 'cl_step_envmajor_kernel<20, true>') == (None, None)      # another instantiation
    summary['_kernel']['kernel'] = 'void (anonymous namespace)::cl_step_envmajor_kernel<20, true>((anonymous namespace)::StepArgs)'
    (prof / f'{cur}b_streaming_pmc_summary.json').write_text(json.dumps(summary))
    traffic, source = bench._pmc_traffic('r*_streaming_pmc_summary.json', 'cl_step_envmajor_kernel<20, true>')
    assert source == f'{cur}b_streaming_pmc_summary.json' and traffic == (2 * 100.0 + 50.0) * 1024.0          # FETCH_SIZE doubled (gfx950), KiB
    line = tmp_path / 'line.json'
    line.write_text(json.dumps({'roofline': {'kernel': 'cl_step_envmajor_kernel<20, true>'}}))
    check = [sys.executable, str(ROOT / 'scripts' / 'check_profiles.py'), str(line)]
    assert subprocess.run(check + [str(prof / f'{cur}b_streaming_pmc_summary.json')], capture_output=True).returncode == 0
    assert subprocess.run(check + [str(prof / f'{cur}a_streaming_pmc_summary.json')], capture_output=True).returncode == 1
    # a line that cites a summary must carry that summary's traffic -- a summary re-collected after the line was written fails
    for traffic, rc in ((traffic, 0), (traffic * 1.06, 1)):
        line.write_text(json.dumps({'roofline': {'kernel': 'cl_step_envmajor_kernel<20, true>', 'traffic': traffic, 'traffic_source': f'{cur}b_streaming_pmc_summary.json'}}))
        assert subprocess.run(check + [str(prof / f'{cur}b_streaming_pmc_summary.json')], capture_output=True).returncode == rc


def _asymmetric_rank(rank, world, port, q):
    """Rank 1's "RCCL" fails at once, rank 0's would sit in its barrier: a gloo subgroup with a short timeout stands in for the RCCL one."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import time
    import torch.distributed as d
    from citylearn_amd.parallel import init_control_plane, reduce_max_seconds
    real_new_group = d.new_group

    def fake_new_group(backend=None, timeout=None, **kw):
        g = real_new_group(backend='gloo', timeout=timeout)
        if rank == 1:
            raise RuntimeError('no RCCL on this rank')
        return g
    d.new_group = fake_new_group
    t0 = time.monotonic()
    cp = init_control_plane(rank, world, None, 'nccl', nccl_timeout_s=5.0, preflight=None)
    took = time.monotonic() - t0
    worst = reduce_max_seconds(float(rank + 1), cp, 'cpu')          # the control plane still works, over gloo
    cp.barrier()
    assert cp.rccl_world_size is None                               # (nothing but gloo carries the control plane now)
    q.put((rank, cp.control_backend, cp.control_fallback, took, worst))
    d.destroy_process_group()


def _preflight_rank(rank, world, port, q):
    """Rank 1 knows locally that it cannot bring RCCL up: nobody may enter an RCCL call (new_group would hang / abort the healthy rank)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as d
    from citylearn_amd.parallel import init_control_plane

    def no_new_group(*a, **kw):
        raise AssertionError('an RCCL communicator was attempted although a rank had failed its pre-flight')
    d.new_group = no_new_group
    cp = init_control_plane(rank, world, None, 'nccl', nccl_timeout_s=5.0, preflight=lambda r, w, dev: 'no xGMI on this rank' if r == 1 else None)
    cp.barrier()
    q.put((rank, cp.control_backend, cp.control_fallback, cp.rccl_world_size))
    d.destroy_process_group()


def test_control_plane_converges_when_rccl_fails_on_one_rank_only():
    """an asymmetric RCCL failure must not strand the healthy ranks.  Every rank joins gloo first and RCCL is a subgroup; the
    ranks agree over gloo whether it came up everywhere -- the failing rank reports at once, the other one after its (short) RCCL timeout,
    and both end on gloo with the reason in `control_fallback`."""
    from citylearn_amd.parallel import free_port
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_asymmetric_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, backend, fallback, took, worst in res:
        assert backend == 'gloo' and fallback and took < 60 and worst == 2.0, res
    assert 'no RCCL on this rank' in res[1][2] and res[0][2]


def test_control_plane_preflight_keeps_every_rank_out_of_rccl():
    """with torch's default async error handling a rank stuck in an RCCL rendezvous is killed by the watchdog, not handed an
    exception -- so a rank that knows locally that RCCL cannot work says so over gloo BEFORE anybody enters an RCCL call."""
    from citylearn_amd.parallel import free_port
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_preflight_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == ['gloo', 'gloo'] and all(r[3] is None for r in res), res
    assert 'no xGMI' in res[1][2] and 'another rank' in res[0][2], res


def test_rank_affinity_arithmetic(tmp_path):
    """`pin_rank_to_gpu_node`'s pieces: cpulist parsing, the per-rank slice of a NUMA node's cores, the fallbacks."""
    from citylearn_amd.parallel import format_cpulist, gpu_numa_node, parse_cpulist, pick_cores
    assert format_cpulist([64, 65, 66, 192, 193, 7]) == '7,64-66,192-193' and parse_cpulist(format_cpulist([3, 4, 9])) == [3, 4, 9]
    assert parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist('') == [] and parse_cpulist('5') == [5]
    node0 = parse_cpulist('0-63,128-191')
    allowed = list(range(256))
    slices = [pick_cores(node0, allowed, 4, i) for i in range(4)]
    assert all(len(s) == 32 for s in slices) and sorted(c for s in slices for c in s) == node0       # disjoint, cover the node
    # every contiguous run is cut separately: a rank gets a core's first hardware thread AND its SMT sibling (c, c + 128)
    assert slices[0] == list(range(16)) + list(range(128, 144)) and slices[3] == list(range(48, 64)) + list(range(176, 192))
    # the affinity mask (a cpuset) cuts the node: slices come from the intersection
    assert pick_cores(node0, list(range(16)), 2, 1) == list(range(8, 16))
    # a cpuset that has (almost) nothing on the GPU's node: the rank keeps every core it was allowed
    assert pick_cores(node0, list(range(64, 80)), 2, 0) == list(range(64, 80))
    assert pick_cores(node0, [0, 1, 2], 2, 1) == [0, 1, 2]
    with pytest.raises(ValueError):
        pick_cores(node0, allowed, 2, 2)
    dev = tmp_path / 'bus' / 'pci' / 'devices' / '0000:c1:00.0'
    dev.mkdir(parents=True)
    (dev / 'numa_node').write_text('1\n')
    assert gpu_numa_node('0000:C1:00.0', str(tmp_path)) == 1 and gpu_numa_node('0000:05:00.0', str(tmp_path)) == -1


def test_shared_device_is_judged_from_the_actual_placement():
    """whether two ranks share a GPU is read off the ranks' (host, physical device) pairs gathered over gloo -- not off a test hook's
    environment variable -- before any rank enters an RCCL call."""
    from citylearn_amd.parallel import shared_device_reason
    assert shared_device_reason([('a', 'uuid::1'), ('a', 'uuid::2'), ('b', 'uuid::1')]) is None            # same UUID on another host: another GPU
    why = shared_device_reason([('a', 'uuid::1'), ('a', 'uuid::2'), ('a', 'uuid::1')])
    assert why is not None and 'ranks 0 and 2' in why and 'RCCL refuses two ranks per device' in why
    assert shared_device_reason([None, ('a', 'index:0')]) is None                                            # a rank without a GPU is the local pre-flight's business
