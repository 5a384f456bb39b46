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
        self.col_src = torch.from_numpy(np.ascontiguousarray(tables.col_src, dtype=np.int32)).to(dev)
        self.col_scale = torch.from_numpy(np.ascontiguousarray(tables.col_scale, dtype=np.float32)).to(dev)
        self.obs = torch.empty((engine.n_env, self.n_cols), dtype=torch.float32, device=dev)
        self.lib.cl_observe_f32.argtypes = [ctypes.POINTER(_lib.Dims)] + [ctypes.c_void_p] * 7 + [ctypes.c_int32] * 3 + [
            ctypes.c_uint32, ctypes.c_void_p]

    def write(self, row: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Observation returned when ``time_step == row`` (row 0: the reset observation, all columns from the table;
        row r >= 1: exogenous values of r + env-dependent values of the step just simulated)."""
        e = self.engine
        out = self.obs if out is None else out
        temp = None if self.stage is None else self.stage.indoor_temp.data_ptr()
        with torch.cuda.device(e.device):
            _lib.check(self.lib.cl_observe_f32(
                ctypes.byref(e.dims), self.table.data_ptr(), self.col_src.data_ptr(), self.col_scale.data_ptr(),
                e.state.data_ptr(), e.out_bldg.data_ptr(), temp, out.data_ptr(), self.n_cols, self.n_rows, int(row),
                abi.CLOB_ALL_EXOGENOUS if row == 0 else 0, torch.cuda.current_stream(e.device).cuda_stream))
        return out

    def algorithmic_bytes(self) -> int:
        """HBM bytes of one launch: the observation tensor written + the dependent planes read (+ one table row)."""
        n_dep = int((self.col_src >= 0).sum().item())
        return 4 * (self.engine.n_env * self.n_cols + self.engine.n_env * n_dep + 3 * self.n_cols)
