"""Device driver of the observation epilogue (`cl_observe_f32`, csrc/cl_observe.h): one launch writes the
``[n_env, n_cols]`` observation tensor from the host-packed tables of :mod:`citylearn_amd.observations` and the
engine's state / output planes."""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib, abi
from .observations import ObservationTables


class ObsDep(ctypes.Structure):
    """`cl_obs_dep` (include/citylearn_amd.h)."""
    _fields_ = [('col', ctypes.c_int32), ('src', ctypes.c_int32), ('scale', ctypes.c_float)]


class ObservationWriter:
    def __init__(self, engine, tables: ObservationTables, stage=None):
        self.lib = _lib.load()
        self.engine, self.stage = engine, stage
        if tables.needs_detail and not (engine.dims.flags & abi.CLD_WRITE_DETAIL):
            raise ValueError('these observations read detail planes: build the StepEngine with detail=True')
        uses_temp = bool(np.any((tables.col_src >= 0) & ((tables.col_src >> 28) == abi.CLOB_KIND_TEMP)))
        if uses_temp and stage is None:
            raise ValueError('indoor-temperature observations need the LSTM stage')
        dev = engine.device
        self.n_rows, self.n_cols = tables.table.shape
        self.table = torch.from_numpy(np.ascontiguousarray(tables.table, dtype=np.float32)).to(dev)
        self.reset_table = None
        if engine.env_row0 is not None:
            if tables.reset_table is None:
                raise ValueError('per-env-block episode windows need ObservationLayout.episode(tables, reset_table=True)')
            self.reset_table = torch.from_numpy(np.ascontiguousarray(tables.reset_table, dtype=np.float32)).to(dev)
        self.col_src = torch.from_numpy(np.ascontiguousarray(tables.col_src, dtype=np.int32)).to(dev)
        self.col_scale = torch.from_numpy(np.ascontiguousarray(tables.col_scale, dtype=np.float32)).to(dev)
        # rows padded to a multiple of 4 floats (16-byte stores); `obs` is the [n_env, n_cols] view of that buffer
        self.pitch = (self.n_cols + 3) // 4 * 4
        self._buffer = torch.zeros((engine.n_env, self.pitch), dtype=torch.float32, device=dev)
        self.obs = self._buffer[:, :self.n_cols]
        # compacted host-side list of the dependent columns: travels in the kernel arguments (fast path)
        cols = np.nonzero(tables.col_src >= 0)[0]
        self.n_deps = len(cols) if len(cols) <= abi.CLOB_MAX_DEPS else -1
        self._deps = None
        if self.n_deps >= 0:
            self._deps = (ObsDep * max(self.n_deps, 1))(*[ObsDep(int(c), int(tables.col_src[c]), float(tables.col_scale[c])) for c in cols])
        self.lib.cl_observe_f32.argtypes = [ctypes.POINTER(_lib.Dims)] + [ctypes.c_void_p] * 4 + [ctypes.c_int32] + [ctypes.c_void_p] * 4 + [
            ctypes.c_int32, ctypes.c_void_p] + [ctypes.c_int32] * 4 + [ctypes.c_uint32, ctypes.c_void_p]
        uses_extra = bool(np.any((tables.col_src >= 0) & ((tables.col_src >> 28) == abi.CLOB_KIND_EXTRA)))
        if uses_extra and engine.flex is None:
            raise ValueError('charging-constraint observations need the flexible-load planes of the engine')

    def write(self, row: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Observation returned when ``time_step == row`` (row 0: the reset observation, all columns from the table;
        row r >= 1: exogenous values of r + env-dependent values of the step just simulated)."""
        e = self.engine
        out = self.obs if out is None else out
        if out.dtype != torch.float32 or out.device != e.device or tuple(out.shape) != (e.n_env, self.n_cols) or out.stride(1) != 1:
            raise ValueError(f'observation buffer must be float32 [{e.n_env}, {self.n_cols}] with unit column stride')
        temp = None if self.stage is None else self.stage.indoor_temp.data_ptr()
        extra, n_extra = (None, 0) if e.flex is None else (e.flex_out.data_ptr(), int(e.flex_out.shape[1]))
        with e._on_device():
            _lib.check(self.lib.cl_observe_f32(
                ctypes.byref(e.dims), (self.reset_table if row == 0 and self.reset_table is not None else self.table).data_ptr(),
                self.col_src.data_ptr(), self.col_scale.data_ptr(),
                ctypes.cast(self._deps, ctypes.c_void_p) if self._deps is not None else None, self.n_deps, e.state.data_ptr(), e.out_bldg.data_ptr(), temp, extra, n_extra, out.data_ptr(), self.n_cols, out.stride(0), self.n_rows, int(row),
                abi.CLOB_ALL_EXOGENOUS if row == 0 else 0, e._stream()))
        return out

    def algorithmic_bytes(self) -> int:
        """HBM bytes of one launch: the observation tensor written + the dependent planes read (+ one table row)."""
        n_dep = int((self.col_src >= 0).sum().item())
        return 4 * (self.engine.n_env * self.n_cols + self.engine.n_env * n_dep + 3 * self.n_cols)
