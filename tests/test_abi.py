"""The C-ABI shared library loads without a GPU and exports every symbol `include/citylearn_amd.h` declares;
argument validation happens before any HIP call, so error codes are testable on CPU."""
import ctypes
from pathlib import Path

import numpy as np
import pytest

from citylearn_amd import _lib, abi


@pytest.fixture(scope='module')
def lib():
    _lib.build()
    return ctypes.CDLL(str(_lib.LIB_PATH))


def test_exports_every_declared_symbol(lib):
    assert abi.EXPORTED_SYMBOLS == ['cl_abi_version', 'cl_finish_f32', 'cl_flex_reset_f32', 'cl_last_error', 'cl_lstm_generic_step_f32', 'cl_lstm_reset_f32', 'cl_lstm_step_f32', 'cl_observe_f32',
                                    'cl_philox_uniform', 'cl_reset_f32', 'cl_rollout_f32', 'cl_rollout_seq_f32', 'cl_step_f32', 'cl_step_flex_f32', 'cl_step_observe_f32']
    for s in abi.EXPORTED_SYMBOLS:
        assert hasattr(lib, s), s
    assert lib.cl_abi_version() == abi.CL_ABI_VERSION


def test_library_exports_nothing_else():
    """The product library exports the declared entry points and nothing more: no tuning hooks, no microbenchmarks
    (those live in libcitylearn_amd_tune.so), no process-global knobs."""
    import subprocess
    _lib.build()
    out = subprocess.run(['nm', '-D', '--defined-only', str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    names = sorted(line.split()[-1] for line in out.splitlines() if ' T ' in line)
    assert names == abi.EXPORTED_SYMBOLS, names


def test_translation_units_and_their_internal_launchers():
    """The library is two translation units: the main one and csrc/cl_noslp_tu.hip (the kernels compiled without SLP vectorisation,
    `-fno-slp-vectorize`); the launchers that connect them are linked in but have hidden visibility -- not part of the C-ABI."""
    import subprocess
    units = [(u, []) if not isinstance(u, tuple) else u for u in _lib.LIB_SOURCES]
    assert [Path(u).name for u, _ in units] == ['cl_kernels.hip', 'cl_noslp_tu.hip']
    assert units[0][1] == [] and units[1][1] == ['-fno-slp-vectorize']
    assert all(Path(u).exists() for u, _ in units)
    _lib.build()
    dyn = subprocess.run(['nm', '-D', '--defined-only', str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    assert 'cl_tu_launch' not in dyn
    full = subprocess.run(['nm', '--defined-only', str(_lib.LIB_PATH)], capture_output=True, text=True).stdout
    if full.strip():                 # (an unstripped build: the launchers are there as local symbols)
        local = {line.split()[-1]: line.split()[-2] for line in full.splitlines() if 'cl_tu_launch' in line}
        assert set(local) == {'cl_tu_launch_rollout', 'cl_tu_launch_lean'} and set(local.values()) == {'t'}, local


def _header_struct_fields(name: str):
    """Field names of `typedef struct <name> {...}` in the header, in declaration order."""
    import re
    text = abi._strip_comments(abi.HEADER.read_text())
    body = re.search(r'typedef\s+struct\s+' + name + r'\s*\{(.*?)\}\s*' + name + r'\s*;', text, flags=re.S).group(1)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        first, *more = decl.split(',')
        for part in [first.split()[-1], *more]:
            fields.append(re.sub(r'\[.*\]|\*', '', part).strip())
    return fields


def _doc_struct_fields(cls: str):
    """Field names of the ctypes stub `class <cls>(ctypes.Structure)` shown in INTEGRATION.md (executed, not pattern-matched)."""
    import re
    from pathlib import Path
    text = (Path(abi.HEADER).parent.parent / 'INTEGRATION.md').read_text()
    m = re.search(r'^class ' + cls + r'\(ctypes\.Structure\):.*?\n(    _fields_ = \[.*?\])[^\n\]]*\n(?=\S|\n)', text, flags=re.S | re.M)
    ns = {'ctypes': ctypes}
    exec('class S(ctypes.Structure):\n' + m.group(1), ns)      # noqa: S102 - our own documentation snippet
    return ns['S']


@pytest.mark.parametrize('c_name,binding,doc', [('cl_dims', _lib.Dims, '_Dims'), ('cl_flex', _lib.Flex, '_Flex'), ('cl_tuning', _lib.Tuning, None)])
def test_struct_layouts_match_the_header(c_name, binding, doc):
    """Field order of the ctypes bindings -- the package's and the stub INTEGRATION.md shows a reference maintainer --
    against the header's struct declarations (a missing field shifts every later pointer)."""
    want = _header_struct_fields(c_name)
    assert [f for f, *_ in binding._fields_] == want
    if doc:
        stub = _doc_struct_fields(doc)
        assert [f for f, *_ in stub._fields_] == want
        assert ctypes.sizeof(stub) == ctypes.sizeof(binding)
        for (fa, ta, *_), (fb, tb, *_) in zip(stub._fields_, binding._fields_):
            assert ctypes.sizeof(ta) == ctypes.sizeof(tb), (fa, fb)


def test_integration_stub_allocates_what_the_header_says():
    """The plane counts the INTEGRATION.md stub allocates with (a maintainer copies them) against the header: a short `out_bldg` is an
    out-of-bounds write for every district that uses the scratch plane (the stub said 15, CL_NO was 18)."""
    import re
    from pathlib import Path
    text = (Path(abi.HEADER).parent.parent / 'INTEGRATION.md').read_text()
    line = re.search(r'^CL_NS, CL_NO, CL_NQ = [^#\n]+', text, flags=re.M).group(0)
    ns = {}
    exec(line, ns)                                          # noqa: S102 - our own documentation snippet
    assert (ns['CL_NS'], ns['CL_NO'], ns['CL_NQ']) == (abi.CL_NS, abi.CL_NO, abi.CL_NQ)
    assert set(re.findall(r'torch\.zeros\(\((CL_\w+),', text)) == {'CL_NS', 'CL_NO', 'CL_NQ'}      # no literal plane counts left in the stub
    assert not re.search(r'torch\.zeros\(\(\d+, B', text)


def test_header_constants_are_consistent():
    assert abi.CLP_USED <= abi.CL_NP and abi.CLP_L_FIRST % 16 == 0 and abi.CLP_L_LAST - abi.CLP_L_FIRST < 32
    assert abi.CLT_ICOP_DHW < abi.CL_NF and abi.CLO_RESERVED < abi.CL_NO and abi.CLQ_REWARD < abi.CL_NQ
    assert ctypes.sizeof(_lib.Dims) == 56 and ctypes.sizeof(_lib.Tuning) == 64          # (56 since ABI 7: env_pitch + a reserved word)
    assert abi.CLD_REWARD_MASK >> abi.CLD_REWARD_SHIFT >= abi.CLR_SOLAR_PENALTY


def test_argument_validation_without_gpu(lib):
    lib.cl_last_error.restype = ctypes.c_char_p
    vp = ctypes.c_void_p
    lib.cl_step_f32.argtypes = [ctypes.POINTER(_lib.Dims), vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp, vp, vp, vp,
                                ctypes.c_int32, vp]
    lib.cl_reset_f32.argtypes = [ctypes.POINTER(_lib.Dims), vp, vp, vp, vp, vp]
    buf = np.zeros(64, dtype=np.float32)
    p = buf.ctypes.data_as(vp)
    assert lib.cl_step_f32(None, p, p, p, p, 4, 1, p, p, None, None, 0, None) == abi.CL_ENULL
    d = _lib.Dims(6, 1, 10, 1, 0)
    assert lib.cl_step_f32(ctypes.byref(d), p, p, p, p, 8, 1, p, p, None, None, 0, None) == abi.CL_EALIGN   # n_env % 4
    assert b'multiple of 4' in lib.cl_last_error()
    d = _lib.Dims(8, 1, 10, 1, 0)
    assert lib.cl_step_f32(ctypes.byref(d), None, p, p, p, 8, 1, p, p, None, None, 0, None) == abi.CL_ENULL
    assert lib.cl_step_f32(ctypes.byref(d), p, p, p, p, 8, 1, p, p, None, None, 10, None) == abi.CL_ERANGE   # t >= n_steps
    assert lib.cl_step_f32(ctypes.byref(d), p, p, ctypes.c_void_p(buf.ctypes.data + 4), p, 8, 1, p, p, None, None, 0, None) == abi.CL_EALIGN
    d = _lib.Dims(8, 1, 10, 1, 9 << abi.CLD_REWARD_SHIFT)
    assert lib.cl_step_f32(ctypes.byref(d), p, p, p, p, 8, 1, p, p, None, None, 0, None) == abi.CL_EINVAL    # unknown reward kind
    d = _lib.Dims(0, 1, 10, 1, 0)
    assert lib.cl_reset_f32(ctypes.byref(d), p, p, None, None, None) == abi.CL_EINVAL
    # cl_step_observe_f32 takes the compact observation form only: every column listed
    lib.cl_step_observe_f32.argtypes = [ctypes.POINTER(_lib.Dims), vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp, vp, vp, vp, ctypes.c_int32,
                                        vp, vp, vp, vp, ctypes.c_int32, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    d = _lib.Dims(8, 1, 10, 1, 0)
    assert lib.cl_step_observe_f32(ctypes.byref(d), p, p, p, p, 8, 1, p, p, None, None, 0, p, p, p, None, 2, p, 2, 4, 10, 1, None) == abi.CL_EINVAL
    assert b'compact form' in lib.cl_last_error()
    assert lib.cl_step_observe_f32(ctypes.byref(d), p, p, p, p, 8, 1, p, p, None, None, 0, None, p, p, p, 2, p, 2, 4, 10, 1, None) == abi.CL_ENULL


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from golden_util import golden
    from citylearn_amd.engine import StepEngine
    tab = golden('g2022_all').spec().episode_tables(0)
    with pytest.raises(_lib.EngineUnavailable):
        StepEngine(tab, 64)


def test_philox_known_answer(lib):
    """Philox4x32-10 known-answer vector of Random123 (counter 0, key 0 -> first word 0x6627e8d5)."""
    lib.cl_philox_uniform.restype = ctypes.c_float
    lib.cl_philox_uniform.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    assert lib.cl_philox_uniform(0, 0, 0, 0) == np.float32((0x6627e8d5 >> 8) / 16777216.0)
    u = np.array([lib.cl_philox_uniform(7, e, 3, 11) for e in range(4096)])
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02
