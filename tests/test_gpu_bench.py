"""bench.py on the GPU box: the multi-rank path as far as one GPU allows (two ranks sharing device 0 through the
CL_BENCH_OVERSUBSCRIBE hook), and one short line per BASELINE config."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _bench(*argv, env=None, timeout=900):
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    e.update(env or {})
    p = subprocess.run([sys.executable, str(ROOT / 'bench.py'), *argv], env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_without_a_launcher():
    """`python bench.py --gpus 2` (no torchrun) starts its own ranks; with the hook both land on device 0.  The N = 1 kernel time
    must be what one rank of the pair sees when it has the GPU to itself (the launches of two processes interleave on one GPU, so
    only the plumbing is asserted on the pair: ranks, world size, per-rank timings, one line)."""
    two = _bench('--gpus', '2', '--steps', '20', '--warmup', '5', '--reps', '3', '--no-streaming', env={'CL_BENCH_OVERSUBSCRIBE': '1'})
    assert two['ranks'] == 2 and two['world_size_seen'] == 2 and two['oversubscribed'] is True and two['n_gpus'] == 1
    assert len(two['rank_ms_per_step']) == 2 and two['control_backend'] == 'gloo'
    assert two['ms_per_step'] >= max(two['rank_ms_per_step']) * 0.5 and two['value'] > 1e9
    assert 'cl_step_lean_chain_kernel<4, true>' in two['roofline']['kernel']                   # the default precision model (CLD_F64_CHAIN)
    one = _bench('--steps', '20', '--warmup', '5', '--reps', '3', '--no-streaming', '--no-cpu-baseline', '--no-traffic-pass')
    assert one['ranks'] == 1 and one['n_gpus'] == 1 and 'oversubscribed' not in one
    assert one['roofline']['kernel'] == 'cl_step_lean_chain_kernel<4, true>' and 0.3 < one['roofline']['frac'] < 1.0
    assert one['value'] == pytest.approx(17 * 65536 / (one['ms_per_step'] * 1e-3))
    assert 'f64' in one['dtype'] and 'CLD_F64_CHAIN' in one['config']['precision']
    # the all-fp32 map rides along as the side entry (the opt-in throughput mode), faster than the default
    side = one['roofline']['fp32_map']['metric_shape']
    assert side['kernel'] == 'cl_step_lean_kernel<4, false, true>' and 1.0 < side['speedup_vs_default'] < 1.5, side
    fast = _bench('--precision', 'fp32', '--steps', '20', '--warmup', '5', '--reps', '2', '--no-streaming', '--no-cpu-baseline', '--no-traffic-pass')
    assert fast['roofline']['kernel'] == 'cl_step_lean_kernel<4, false, true>' and fast['dtype'] == 'f32' and 'fp32_map' not in fast['roofline']


def test_headline_line_measures_its_hbm_traffic_in_the_run():
    """`roofline.traffic` of the default line is MEASURED by the run itself (two `rocprofv3 --pmc` child passes, FETCH_SIZE / WRITE_SIZE,
    counter collection only), not read from a committed file: it must land on the algorithmic byte count (no wasted re-reads)."""
    import shutil
    if shutil.which('rocprofv3') is None and not Path('/opt/rocm/bin/rocprofv3').exists():
        pytest.skip('rocprofv3 not installed')
    out = _bench('--steps', '20', '--warmup', '5', '--reps', '2', '--no-streaming', '--no-cpu-baseline')
    r = out['roofline']
    assert 'traffic_live_error' not in r, r.get('traffic_live_error')
    assert r['traffic_source'].startswith('measured in this run') and 'cl_step_lean_chain_kernel<4, true>' in r['traffic_source']
    algorithmic = r['algorithmic_bytes_per_unit'] * r['units_per_launch']
    assert 0.97 < r['traffic'] / algorithmic < 1.10, (r['traffic'], algorithmic)


def test_dry_run_is_refused_where_a_gpu_is_visible():
    """CL_BENCH_DRY_RUN prints a complete line of made-up timings (CPU launcher tests): a box that can measure must never emit one."""
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    e['CL_BENCH_DRY_RUN'] = '1'
    p = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--steps', '5', '--warmup', '2'], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and 'CL_BENCH_DRY_RUN' in p.stderr and not p.stdout.strip()


def test_scale_run_extras_on_the_one_gpu_there_is():
    """What only the driver's N > 1 run executes otherwise: the C4 / C4-lean / C5 lines measured in the same run
    (`extra_configs`), per-rank kernel time next to per-rank wall time, and the ranks' CPU pinning -- two ranks sharing device 0."""
    two = _bench('--gpus', '2', '--steps', '20', '--warmup', '5', '--reps', '2', '--no-streaming', env={'CL_BENCH_OVERSUBSCRIBE': '1', 'CL_BENCH_EXTRA_CONFIGS': '1'})
    assert len(two['rank_launch_us']) == 2 and all(3.0 < k < 60.0 for k in two['rank_launch_us']), two['rank_launch_us']
    assert set(two['extra_configs']) == {'fixed-65536', 'C4', 'C4-lean', 'C5', 'C4-B', 'C4-lean-B'}
    for name, x in two['extra_configs'].items():
        assert x['value'] > 1e9 and len(x['rank_ms_per_step']) == 2 and len(x['rank_launch_us']) == 2 and x['roofline']['kernel'], name
    fx = two['extra_configs']['fixed-65536']                                               # ONE 65 536-env batch over the ranks: north_star's other reading
    assert fx['scaling'] == 'strong' and fx['envs_per_gpu'] == 32768 and 'chain' in fx['roofline']['kernel']
    assert 'cl_rollout_kernel' in two['extra_configs']['C5']['roofline']['kernel']
    assert two['extra_configs']['C4-B']['roofline']['kernel'].startswith('cl_rollout_full_kernel<1, true, 2, false>')      # (config 4's thermal district in mode B: the packed unit)
    assert two['extra_configs']['C4-B']['value'] > two['extra_configs']['C4']['value']
    aff = two['rank_affinity']
    assert len(aff) == 2
    if aff[0] is not None:                                   # (None: the box does not expose the GPU's NUMA node)
        assert aff[0]['ranks_on_node'] == 2 and aff[0]['n_cores'] >= 1 and aff[0]['cores'] != aff[1]['cores']


def test_more_ranks_than_gpus_is_refused_without_the_hook():
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'CL_BENCH_OVERSUBSCRIBE')}
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', str(n), '--steps', '5', '--warmup', '2'], env=e, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode != 0 and 'CL_BENCH_OVERSUBSCRIBE' in p.stderr and not p.stdout.strip()


@pytest.mark.parametrize('cfg,kernel,bound,prec', [('C2', 'cl_step_lean_chain_kernel<1, true>', 'hbm', 'chain'), ('C2', 'cl_step_lean_kernel<1, false, true>', 'hbm', 'fp32'),
                                                   ('C3', 'cl_lstm_kernel<', 'valu', 'chain'), ('C4', 'cl_step_full_', 'hbm', 'chain'),
                                                   ('C4', 'cl_step_full_kernel<2, false, 1024, 4, true, true>', 'hbm', 'fp32'),
                                                   ('C5', 'cl_rollout_kernel<2, false, 2, true, false, 2>', 'valu', 'chain'),
                                                   ('C5', 'cl_rollout_kernel<2, false, 2, true, false, 0>', 'valu', 'fp32'),
                                                   ('T9', 'cl_step_full_tp_chain_kernel<1, 4, true>', 'hbm', 'chain')])
def test_config_lines(cfg, kernel, bound, prec):
    out = _bench('--config', cfg, '--precision', prec, '--steps', '20', '--warmup', '5', '--reps', '2')
    assert out['config']['name'] == cfg and out['roofline']['bound'] == bound and kernel in out['roofline']['kernel'], out['roofline']['kernel']
    assert out['value'] > 1e8 and (out['roofline']['frac'] is None or 0.0 < out['roofline']['frac'] < 1.0)       # (the chain's fused rollout claims no VALU fraction)
    if cfg == 'C2':
        # config 2 is launch latency in mode A (2.7 MB per step): the line says what mode B gives a user at the same batch size
        mb = out['roofline']['mode_b']
        assert 'cl_rollout_kernel' in mb['kernel'] and mb['speedup_vs_mode_a'] > 1.5 and mb['value'] > out['value'], mb
    if cfg == 'T9':
        # ... and the thermal district in mode B: the packed unit inside the K-step loop (round 6)
        mb = out['roofline']['mode_b']
        assert mb['kernel'].startswith('cl_rollout_full_kernel<1, false, 2, false>') and mb['speedup_vs_mode_a'] > 1.0, mb


def test_thermal_kpi_line_runs_the_kpis_inside_the_step_launch():
    """`bench.py --config T9 --kpi`: one launch per step (`cl_step_full_kpi_kernel`), priced against HBM with the ten per-unit accumulators
    and the moving district-series values in the byte count."""
    out = _bench('--config', 'T9', '--kpi', '--precision', 'fp32', '--steps', '20', '--warmup', '5', '--reps', '2', '--no-cpu-baseline')
    r = out['roofline']
    assert out['config']['name'] == 'T9' and r['bound'] == 'hbm' and r['kernel'] == 'cl_step_full_kpi_kernel<true>', r['kernel']
    assert 140.0 < r['algorithmic_bytes_per_unit'] < 160.0 and 0.3 < r['frac'] < 1.0


def test_rccl_control_plane_keeps_stdout_to_the_one_line():
    """A lone rank with the process group forced up (CL_BENCH_FORCE_DIST): the control plane is RCCL -- the communicator really is
    created on this GPU -- and RCCL's banner, which goes through buffered C stdio, does not end up on stdout behind the JSON line."""
    out = _bench('--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-streaming', '--no-traffic-pass', env={'CL_BENCH_FORCE_DIST': '1'})      # (_bench asserts the single line)
    # (RCCL came up on every box of round 3; should a box refuse it, the fallback must have taken over -- still one line)
    assert out['world_size_seen'] == 1 and (out['control_backend'] == 'nccl' or out.get('control_fallback'))


def test_control_plane_falls_back_to_gloo_when_rccl_refuses():
    """Two ranks on ONE device with RCCL forced (it refuses: "Duplicate GPU detected", in every rank): the barrier and the MAX over ranks
    move to gloo, the run completes, and the line says why."""
    out = _bench('--gpus', '2', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-streaming',
                 env={'CL_BENCH_OVERSUBSCRIBE': '1', 'CL_BENCH_CONTROL': 'nccl'})
    assert out['ranks'] == 2 and out['world_size_seen'] == 2 and out['control_backend'] == 'gloo'
    # (round 5: the ranks learn it from the local pre-flight -- two ranks on one device -- before anybody enters an RCCL call; RCCL's own
    #  "Duplicate GPU detected" is what the communicator probe would have raised)
    assert any(k in out['control_fallback'] for k in ('Duplicate GPU', 'NCCL', 'RCCL refuses two ranks per device')), out['control_fallback']
    assert out['rccl_world_size'] is None
    assert len(out['rank_ms_per_step']) == 2


def test_default_line_carries_the_hbm_true_roofline_and_the_dropin_timing():
    """The line the driver records: `roofline.frac` is the HBM-TRUE fraction -- the step kernel at 17 x 1 048 576 envs,
    measured in the same run -- with the cache-resident launch the value is timed on as `roofline.metric_shape`; `cpu_baseline` is the reference's
    own step over >= 1000 steps per process; `dropin` times `citylearn_amd.CityLearnEnv` over config 1's full episode beside the reference's."""
    out = _bench('--steps', '20', '--warmup', '5', '--reps', '2', '--no-traffic-pass', timeout=1500)
    r = out['roofline']
    assert r['bound'] == 'hbm' and r['kernel'] == 'cl_step_lean_chain_kernel<4, true>' and r['units_per_launch'] == 17 * 1048576      # (round 6: the building-major kernel from 16 Mi units)
    assert 0.3 < r['frac'] < 0.9 and r['frac'] == pytest.approx(r['achieved'] / 8000.0)
    m = r['metric_shape']
    assert m['kernel'] == 'cl_step_lean_chain_kernel<4, true>' and 'infinity-cache' in m['residency'] and m['units_per_launch'] == 17 * 65536
    assert out['value'] == pytest.approx(17 * 65536 / (out['ms_per_step'] * 1e-3))
    assert r['fp32_map']['hbm_streaming']['kernel'].startswith('cl_step_lean_kernel<4, ') and r['fp32_map']['hbm_streaming']['speedup_vs_default'] > 0.9
    cb = out['cpu_baseline']
    if cb.get('kind') == 'reference' and 'live' in cb.get('measured', ''):
        assert cb['all_cores']['steps'] >= 1000
    d = out['dropin']
    assert 'error' not in d, d
    assert d['steps'] == 8759 and d['value'] > 5 * 8759 / 60.0           # the year in well under a minute
