"""Static guards on the generated gfx950 code (no GPU: hipcc cross-compiles to assembly).  Each assertion pins a property that cost
a measured factor when the compiler silently lost it: wave-uniform parameter reads as SCALAR loads, no scratch
memory in the rollout / lean / LSTM kernels, the LSTM window loop free of a load in front of its use, no packed fp32 in the
kernels that were moved to the no-SLP translation unit."""
import os
import re
import subprocess

import pytest

from citylearn_amd import _lib


def _asm(src, extra, tmp_path_factory):
    out = tmp_path_factory.mktemp('isa') / (src.stem + '.s')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    flags = [f for f in _lib.HIPCC_FLAGS if f not in ('-shared', '-fPIC')]
    subprocess.run([hipcc, *flags, *extra, '-S', '--cuda-device-only', str(src), '-o', str(out)], check=True, capture_output=True)
    kernels, meta, cur = {}, {}, None
    for line in out.read_text().splitlines():
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1); kernels[cur] = []
            continue
        m = re.match(r'\s*\.set (_Z\w+)\.(num_vgpr|private_seg_size), (\d+)', line)
        if m:
            meta.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
            continue
        t = line.split()
        if cur is not None and re.match(r'^\.LBB\d+_\d+:', line):
            kernels[cur].append('LABEL ' + line.split(':')[0])
        elif cur is not None and t and not t[0].startswith(('.', ';')):
            kernels[cur].append(line.strip())
    return kernels, meta


@pytest.fixture(scope='module')
def units(tmp_path_factory):
    (main, _), (noslp, noslp_flags) = [(u, []) if not isinstance(u, tuple) else u for u in _lib.LIB_SOURCES]
    return _asm(main, [], tmp_path_factory), _asm(noslp, noslp_flags, tmp_path_factory)


def _one(kernels, pattern):
    names = [k for k in kernels if re.search(pattern, k)]
    assert len(names) == 1, (pattern, names)
    return names[0]


def _count(ins, prefix):
    return sum(1 for i in ins if i.split()[0].startswith(prefix))       # ('LABEL' pseudo-entries never match an opcode prefix)


def test_headline_lean_kernel(units):
    """cl_step_lean_kernel<4, false, true> (17 x 65 536): parameters and time-series rows through scalar loads (the traced build that
    lost them ran 3.5 x slower), no scratch, no packed fp32 (same-box A/B: 7.17 -> 6.86 us without)."""
    _, (kernels, meta) = units
    k = _one(kernels, r'cl_step_lean_kernelILi4ELb0ELb1EE')
    ins = kernels[k]
    assert meta[k]['private_seg_size'] == 0
    assert _count(ins, 'global_load') <= 32 and _count(ins, 's_load') >= 15, (_count(ins, 'global_load'), _count(ins, 's_load'))
    assert not [i for i in ins if re.match(r'v_pk_\w+_f32', i)]
    assert meta[k]['num_vgpr'] <= 96                       # five waves per SIMD and one 9-wave workgroup per CU


def test_rollout_kernels_have_no_scratch_and_no_packed_fp32(units):
    """The Philox block cache once went through scratch memory (a dynamically indexed struct): 16-byte scratch store + load per env,
    building and step; packed fp32 costs the two-envs-per-lane kernel 9 %."""
    _, (kernels, meta) = units
    names = [k for k in kernels if 'cl_rollout_kernel' in k]
    assert len(names) >= 5
    for k in names:
        # (the chunked two-env chain instantiation reserves 20 bytes it never touches: no scratch instruction in its body)
        assert meta[k]['private_seg_size'] == (20 if 'ILi2ELb0ELi2ELb1ELb1ELi2E' in k else 0), k
        assert not [i for i in kernels[k] if i.startswith('scratch_') or re.match(r'v_pk_\w+_f32', i)], k


def test_main_unit_no_longer_holds_the_moved_kernels(units):
    (kernels, _), _ = units
    assert not [k for k in kernels if 'cl_rollout_kernel' in k or re.search(r'cl_step_lean_kernelILi\dELb0E', k)]


def test_thermal_kernel_parameters_are_scalar_loads(units):
    """cl_step_full_tp_kernel<1, 4, *> (2020 schema 9 x 65 536): the parameter block of a (tile, building) item is read with s_load."""
    (kernels, meta), _ = units
    k = _one(kernels, r'cl_step_full_tp_kernelILi1ELi4ELb1EE')
    assert _count(kernels[k], 's_load') >= 30 and _count(kernels[k], 'global_load') <= 16
    assert meta[k]['private_seg_size'] == 0


def test_lstm_window_loop(units):
    """cl_lstm_kernel<0, 2> (f16 split): two waves per SIMD (<= 256 registers, no scratch); the window loop issues its three input
    loads at the top and waits for them only at its end -- the compiler once sank the history load right in front of its use (one
    exposed memory latency per window step: 176 vs 150 us)."""
    (kernels, meta), _ = units
    # the both-demand instantiation (CLD_LSTM_TWO_DEMANDS): one more load per window step, the same register budget
    k2 = _one(kernels, r'cl_lstm_kernelILi0ELi2ELb1EE')
    assert meta[k2]['num_vgpr'] <= 256 and meta[k2]['private_seg_size'] == 0
    k = _one(kernels, r'cl_lstm_kernelILi0ELi2ELb0EE')
    assert meta[k]['num_vgpr'] <= 256 and meta[k]['private_seg_size'] == 0
    ins = kernels[k]
    # the loop: a backward conditional branch to a label, with the 18 f16 MFMAs of a window step in between
    labels = {x.split()[1]: i for i, x in enumerate(ins) if x.startswith('LABEL ')}
    loops = [(labels[x.split()[-1]], i) for i, x in enumerate(ins)
             if x.startswith('s_cbranch') and x.split()[-1] in labels and labels[x.split()[-1]] < i]
    loops = [(a, b) for a, b in loops if sum('v_mfma_f32_32x32x16_f16' in x for x in ins[a:b]) == 18]
    assert len(loops) == 1, loops
    a, b = loops[0]
    body = [x for x in ins[a + 1:b] if not x.startswith('LABEL ')]
    loads = [i for i, x in enumerate(body) if x.startswith('global_load')]
    assert len(loads) == 3 and loads[-1] < 60, loads                         # all three at the top
    waits0 = [i for i, x in enumerate(body) if x.startswith('s_waitcnt') and 'vmcnt(0)' in x]
    assert waits0 and waits0[0] > len(body) - 40, (waits0, len(body))         # the only full wait: the rotation at the very end
    assert sum('v_exp_f32' in x for x in body) == 80 and sum('v_rcp_f32' in x for x in body) == 80      # 10 per hidden unit and cell


def test_f64_map_kernels_have_no_scratch(units):
    """CLD_F64_MAPS instantiations: the float64 battery parameters selected by value, not through struct addresses (the first build
    parked the 26-double block in scratch: 208 bytes per lane), and the thermal unit only at one env per lane (it spills at two)."""
    (kernels, meta), _ = units
    names = [k for k in kernels if 'cl_step_lean_f64_kernel' in k or re.search(r'cl_step_kernelILi[12]ELb[01]ELb[01]ELb0ELi1ELb0ELb0EE', k)]
    assert len(names) >= 8, names
    for k in names:
        assert meta[k]['private_seg_size'] == 0, k
        assert any(i.startswith('v_div_scale_f64') or i.startswith('v_rcp_f64') for i in kernels[k]), k      # true float64 divisions, like the reference
        assert any(i.startswith('v_fma_f64') or i.startswith('v_mul_f64') for i in kernels[k]), k


def test_in_kernel_fold_uses_scoped_accesses_not_cache_maintenance(units):
    """`district_reduce<.., FOLD>` (cl_tuning.finish = 2): the chunk partial sums and the ticket cross XCDs through agent-scope accesses
    (`sc1` stores / loads, one atomic) -- and nothing in the kernel writes back or invalidates the L2 (round 1's version did: 177 us)."""
    (kernels, _), _ = units
    for pat in (r'cl_step_full_kernelILi2ELb0ELi1024ELi4ELb1ELb1EE', r'cl_step_kernelILi4ELb0ELb0ELb0ELi0ELb1ELb0EE'):
        ins = kernels[_one(kernels, pat)]
        assert not [i for i in ins if i.startswith(('buffer_wbl2', 'buffer_inv'))]
        assert sum(1 for i in ins if i.startswith('global_store_dword') and i.endswith('sc1')) >= 2        # partial sums, ticket reset
        assert sum(1 for i in ins if i.startswith('global_load_dword') and i.endswith('sc1')) >= 16        # the fold: sixteen loads in flight
        assert sum(1 for i in ins if i.startswith('global_atomic_add')) == 1
    # the instantiations that never fold keep their register budget (the fold is a template parameter, not a run-time branch)
    k = _one(kernels, r'cl_step_kernelILi1ELb0ELb0ELb0ELi0ELb0ELb0EE')
    assert not [i for i in kernels[k] if i.endswith('sc1')]


def test_scratch_memory_is_confined_to_the_known_instantiations(units):
    """No kernel of the library touches scratch memory except the ones listed here with their byte counts (two or three VGPRs parked once per
    wave under a 1024-thread workgroup's 128-register cap; the FLEX thermal kernel with detail planes: 13): a new entry is a regression."""
    known = {r'cl_step_kernelILi2ELb1ELb1ELb1ELi0ELb0ELb0EE': 52, r'cl_step_full_kernelILi2ELb1ELi1024ELi4ELb0ELb[01]EE': 8,
             r'cl_step_full_kernelILi2ELb0ELi576ELi5ELb0ELb[01]EE': 8, r'cl_step_full_kernelILi2ELb0ELi1024ELi5ELb0ELb[01]EE': 8,     # (the second: forced launches only)
             # the thermal step with the streaming KPI epilogue: 36 bytes RESERVED (slots of scalar registers that ended up parked in
             # vector-register lanes instead) and never accessed -- checked below
             r'cl_step_full_kpi_kernelILb[01]E': 36,
             # CLD_CHECK (round 6): the DEBUG instantiations of the general kernel -- one violation word more per unit; a few registers parked in
             # scratch cost a 4-env single-district launch nothing (never selected for a production batch)
             r'cl_step_kernelILi1ELb1ELb1ELb[01]ELi[012]ELb0ELb1EE': 64,
             # the battery + PV chunk kernel around the float64 chain WITH the deferred fold at four envs per lane: three registers parked once per wave
             r'cl_step_lean_chunk_kernelILi4ELb[01]ELb1ELi2EE': 12,
             # the chunked fused rollout around the chain at two envs per lane: 20 bytes RESERVED and never accessed (test_rollout_kernels_have_no_scratch_...)
             r'cl_rollout_kernelILi2ELb0ELi2ELb1ELb1ELi2EE': 20}
    # the building-chunked thermal launches (BASELINE config 4; parameter blocks staged in LDS): 16 / 12 bytes per lane until round 4 -- the C4
    # shard's 1.145 x HBM traffic -- none since their district accumulators live in the wave's LDS row (cl_full.h, QLDS)
    main_meta = units[0][1]
    for pat in (r'cl_step_full_kernelILi2ELb0ELi1024ELi4ELb1ELb[01]EE', r'cl_step_full_kernelILi1ELb0ELi1024ELi5ELb1ELb[01]EE'):
        hits = [k for k in main_meta if re.search(pat, k)]
        assert len(hits) == 2 and all(main_meta[k]['private_seg_size'] == 0 for k in hits), (pat, [(k, main_meta[k]) for k in hits])
    # the latency-ordered chunk kernel of battery + PV districts (round 5): the next building's inputs reuse the registers the arithmetic
    # released -- a second input set or LDS-staged blocks cost 36 - 44 bytes of scratch at four envs per lane
    hits = [k for k in main_meta if re.search(r'cl_step_lean_chunk_kernelILi4ELb[01]ELb[01]ELi0EE', k)]
    assert len(hits) == 4 and all(main_meta[k]['private_seg_size'] == 0 and main_meta[k]['num_vgpr'] <= 128 for k in hits), [(k, main_meta[k]) for k in hits]
    # ... and around the float64 soc chain (round 6, PREC = 2): clean without the fold; the folding instantiation parks three registers (`known`)
    hits = [k for k in main_meta if re.search(r'cl_step_lean_chunk_kernelILi[14]ELb[01]ELb0ELi2EE', k)]
    assert len(hits) == 4 and all(main_meta[k]['private_seg_size'] == 0 and main_meta[k]['num_vgpr'] <= 128 for k in hits), [(k, main_meta[k]) for k in hits]
    for kernels, meta in units:
        for k, m in meta.items():
            if not m.get('private_seg_size'):
                continue
            hit = [limit for pat, limit in known.items() if re.search(pat, k)]
            assert hit and m['private_seg_size'] <= hit[0], (k, m['private_seg_size'])
            if 'cl_step_full_kpi_kernel' in k:
                assert m['num_vgpr'] <= 128 and not [i for i in kernels[k] if i.startswith(('scratch_', 'buffer_load', 'buffer_store'))], k
