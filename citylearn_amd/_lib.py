"""ctypes binding of ``libcitylearn_amd.so`` (the C-ABI in ``include/citylearn_amd.h``).

There is deliberately NO fallback: if the shared library is missing or a call fails the error propagates
(``EngineUnavailable`` / ``EngineError``).  Build with ``python -c 'import __graft_entry__ as g; g.build()'``
(or ``make -C citylearn_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

from . import abi

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / 'libcitylearn_amd.so'
TUNE_LIB_PATH = PKG / 'libcitylearn_amd_tune.so'
CSRC = PKG / 'csrc'
# -amdgpu-mfma-vgpr-form: MFMA accumulators stay in VGPRs (gfx950's register file is unified), which removes the
# v_accvgpr_read copies in front of the LSTM activations (64 per window step)
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-mllvm', '-amdgpu-mfma-vgpr-form']


class EngineUnavailable(RuntimeError):
    pass


class EngineError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f'citylearn_amd C-ABI error {code}: {message}')
        self.code = code


class Tuning(ctypes.Structure):
    """``cl_tuning`` (include/citylearn_amd.h): per-call launch-geometry overrides; all zero = library defaults."""
    _fields_ = [('vec', ctypes.c_int32), ('nw', ctypes.c_int32), ('no_chunks', ctypes.c_int32), ('lean_variant', ctypes.c_int32),
                ('envmajor', ctypes.c_int32), ('flex_vec', ctypes.c_int32), ('obs_variant', ctypes.c_int32),
                ('obs_rows', ctypes.c_int32), ('lstm_variant', ctypes.c_int32), ('full_variant', ctypes.c_int32),
                ('b_chunk', ctypes.c_int32), ('nt_stores', ctypes.c_int32), ('kernel_name', ctypes.c_void_p),
                ('finish', ctypes.c_int32), ('kpi_passes', ctypes.c_int32)]


class Dims(ctypes.Structure):
    """``cl_dims`` (include/citylearn_amd.h)."""
    _fields_ = [('n_env', ctypes.c_int32), ('n_bldg', ctypes.c_int32), ('n_steps', ctypes.c_int32),
                ('n_act_cols', ctypes.c_int32), ('flags', ctypes.c_uint32), ('n_ts_rows', ctypes.c_int32),
                ('env_row0', ctypes.c_void_p), ('tuning', ctypes.POINTER(Tuning)), ('env_offset', ctypes.c_int64),
                ('env_pitch', ctypes.c_int32), ('reserved0', ctypes.c_int32)]


class Flex(ctypes.Structure):
    """``cl_flex`` (include/citylearn_amd.h): EV chargers / washing machines tables and state."""
    _fields_ = [('n_ev', ctypes.c_int32), ('n_flex_bldg', ctypes.c_int32), ('n_rows', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('ev_params', ctypes.c_void_p), ('ev_ts', ctypes.c_void_p), ('charger_params', ctypes.c_void_p),
                ('charger_ts', ctypes.c_void_p), ('wm_params', ctypes.c_void_p), ('wm_ts', ctypes.c_void_p),
                ('cons_params', ctypes.c_void_p),
                ('ev_state', ctypes.c_void_p), ('wm_state', ctypes.c_void_p),
                ('flex_out', ctypes.c_void_p), ('charger_out', ctypes.c_void_p), ('drift', ctypes.c_void_p),
                ('seed', ctypes.c_uint64), ('weights', ctypes.c_float * 8)]


def _compile(sources, out: Path, deps, force: bool, verbose: bool) -> Path:
    """`sources`: paths, or (path, extra flags) pairs -- translation units with flags of their own are compiled to objects first."""
    if not force and out.exists() and all(out.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return out
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    units = [(s, []) if not isinstance(s, tuple) else s for s in sources]

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True)

    if all(not extra for _, extra in units):
        run([hipcc, *HIPCC_FLAGS, *(str(s) for s, _ in units), '-o', str(out)])
        return out
    import shutil
    tmp = out.parent / f'build_{out.stem}_{os.getpid()}'           # objects stay inside the tree (git-ignored) and are removed again
    tmp.mkdir(exist_ok=True)
    try:
        objs = []
        for src, extra in units:
            obj = tmp / (Path(src).stem + '.o')
            run([hipcc, *[f for f in HIPCC_FLAGS if f != '-shared'], *extra, '-c', str(src), '-o', str(obj)])
            objs.append(str(obj))
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', str(out)])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


# cl_noslp_tu.hip: the fused rollout kernel and the plain lean step kernel without SLP vectorisation (packed fp32 operations cost them
# 9 % / 4 %; csrc/cl_kernels.hip)
LIB_SOURCES = [CSRC / 'cl_kernels.hip', (CSRC / 'cl_noslp_tu.hip', ['-fno-slp-vectorize'])]


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP sources for gfx950 into the in-tree shared library (cross-compiles without a GPU)."""
    return _compile(LIB_SOURCES, LIB_PATH, sorted(CSRC.glob('*.hip')) + sorted(CSRC.glob('*.h')) + [abi.HEADER], force, verbose)


def build_variant(out: Path, extra_flags, verbose: bool = False) -> Path:
    """A second build of the library with extra compiler flags on every translation unit (diagnostic / A-B builds of scripts/)."""
    units = [(s, list(extra_flags)) if not isinstance(s, tuple) else (s[0], [*s[1], *extra_flags]) for s in LIB_SOURCES]
    return _compile(units, Path(out), [], True, verbose)


def build_tune(force: bool = False, verbose: bool = False) -> Path:
    """The microbenchmark / layout-probe kernels (csrc/cl_tune.hip) live in their own library, which the package never loads."""
    sources = [CSRC / 'cl_tune.hip']
    return _compile(sources, TUNE_LIB_PATH, sources, force, verbose)


def load_tune() -> ctypes.CDLL:
    """``libcitylearn_amd_tune.so`` for scripts/ and the MFMA layout test (not used by the package)."""
    import torch  # noqa: F401
    if not TUNE_LIB_PATH.exists():
        raise EngineUnavailable(f'{TUNE_LIB_PATH} not found (run __graft_entry__.build())')
    lib = ctypes.CDLL(str(TUNE_LIB_PATH))
    lib.cl_tune_copy_floor.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.cl_tune_mfma_bench.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.cl_tune_mfma_bf16_probe.argtypes = [ctypes.c_void_p] * 4
    return lib


_lib = None


def load() -> ctypes.CDLL:
    """Load the library (after torch, so that both share one libamdhip64 / one HIP context)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- must come first: the HIP runtime torch ships is the one the kernels run on
    # CITYLEARN_AMD_LIB: an alternative build of the same library (A/B builds of scripts/: `build_variant`), never a different backend
    path = Path(os.environ['CITYLEARN_AMD_LIB']).resolve() if os.environ.get('CITYLEARN_AMD_LIB') else LIB_PATH
    if not path.exists():
        raise EngineUnavailable(f'{path} not found: the HIP extension is not built (run __graft_entry__.build())')
    lib = ctypes.CDLL(str(path))
    vp, i32, i64, u64, f32p = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p
    lib.cl_abi_version.restype = ctypes.c_int
    lib.cl_last_error.restype = ctypes.c_char_p
    lib.cl_reset_f32.restype = ctypes.c_int
    lib.cl_reset_f32.argtypes = [ctypes.POINTER(Dims), vp, f32p, f32p, f32p, vp]
    lib.cl_step_f32.restype = ctypes.c_int
    lib.cl_step_f32.argtypes = [ctypes.POINTER(Dims), vp, f32p, f32p, f32p, i64, i64, f32p, f32p, f32p, f32p, i32, vp]
    lib.cl_flex_reset_f32.restype = ctypes.c_int
    lib.cl_flex_reset_f32.argtypes = [ctypes.POINTER(Dims), ctypes.POINTER(Flex), vp]
    lib.cl_step_flex_f32.restype = ctypes.c_int
    lib.cl_step_flex_f32.argtypes = [ctypes.POINTER(Dims), vp, f32p, f32p, f32p, i64, i64, f32p, f32p, f32p, f32p,
                                     ctypes.POINTER(Flex), i32, vp]
    lib.cl_step_observe_f32.restype = ctypes.c_int
    lib.cl_step_observe_f32.argtypes = [ctypes.POINTER(Dims), vp, f32p, f32p, f32p, i64, i64, f32p, f32p, f32p, f32p, i32,
                                        f32p, vp, f32p, vp, i32, f32p, i32, i32, i32, i32, vp]
    lib.cl_lstm_generic_step_f32.restype = ctypes.c_int
    lib.cl_rollout_f32.restype = ctypes.c_int
    lib.cl_rollout_f32.argtypes = [ctypes.POINTER(Dims), vp, f32p, f32p, f32p, i64, i64, i64, f32p, f32p, u64,
                                   f32p, f32p, f32p, i32, i32, vp]
    lib.cl_rollout_seq_f32.restype = ctypes.c_int
    lib.cl_rollout_seq_f32.argtypes = [ctypes.POINTER(Dims), vp, f32p, f32p, f32p, i64, i64, i64, f32p, f32p, u64, f32p,
                                       f32p, f32p, f32p, f32p, f32p, ctypes.POINTER(Flex), i32, i32, vp]
    lib.cl_finish_f32.restype = ctypes.c_int
    lib.cl_finish_f32.argtypes = [ctypes.POINTER(Dims), f32p, f32p, i32, vp]
    lib.cl_philox_uniform.restype = ctypes.c_float
    lib.cl_philox_uniform.argtypes = [u64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    got = lib.cl_abi_version()
    if got != abi.CL_ABI_VERSION:
        raise EngineUnavailable(f'ABI mismatch: library {got}, header {abi.CL_ABI_VERSION}; rebuild the extension')
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise EngineError(rc, load().cl_last_error().decode(errors='replace'))
