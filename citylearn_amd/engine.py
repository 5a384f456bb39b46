"""Device-side step engine: HBM-resident tables + state, advanced by the HIP kernels through the C-ABI.

This is the replacement for the per-building Python loops of ``CityLearnEnv.step`` (reference
citylearn/citylearn.py:1010-1027) for a whole batch of independent environments.  Tensors are PyTorch-ROCm
tensors (plumbing for device memory and streams only); all arithmetic happens in
``citylearn_amd/csrc/cl_kernels.hip``.
"""
from __future__ import annotations

import ctypes
from typing import Mapping, Optional

import numpy as np
import torch

from . import _lib, abi
from .schema import EpisodeTables

REWARD_KINDS = {
    'RewardFunction': abi.CLR_DEFAULT,
    'MARL': abi.CLR_MARL,
    'IndependentSACReward': abi.CLR_INDEPENDENT_SAC,
    'SolarPenaltyReward': abi.CLR_SOLAR_PENALTY,
    'Electric_Vehicles_Reward_Function': abi.CLR_EV,
}


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL_CONTEXT = _NullContext()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class StepEngine:
    """One env shard on one GPU.

    Layouts follow include/citylearn_amd.h: ``state [CL_NS, B, E]``, ``out_bldg [CL_NO, B, E]``,
    ``out_env [CL_NQ, E]``; ``actions`` is ``[n_act_cols, E]`` (coalesced) or any strided 2-D view.
    """

    def __init__(self, tables: EpisodeTables, n_env: int, device: str = 'cuda:0', reward: str = 'RewardFunction',
                 t0_quirk: bool = True, detail=False, n_act_cols: Optional[int] = None, kpi: bool = False,
                 n_steps: Optional[int] = None, env_row0=None, ev_reward_weights=None, ev_drift=None, ev_seed: int = 0,
                 charger_detail: bool = False, ev_penalty_coefficient: float = 1.0, central_agent: bool = False, tuning: Optional[dict] = None,
                 env_offset: int = 0, f64_maps=None, env_pitch: Optional[int] = None, check: bool = False):
        """`n_steps` / `env_row0`: per-env-block episode windows (`cl_dims.env_row0`).  `tables` then covers the whole
        simulation period, an episode is `n_steps` rows long and block g of `abi.CL_ROW0_BLOCK` consecutive envs starts
        at table row ``env_row0[g]`` -- different blocks replay different windows at once.

        Districts with EV chargers / washing machines (``tables.flex``): ``ev_reward_weights`` are the `weights` of
        Electric_Vehicles_Reward_Function, ``ev_seed`` keys the on-device N(1, 0.2) stream of the unconnected-EV SoC
        drift (citylearn.py:1468-1472) and ``ev_drift`` ([table rows, n_ev], optional) replays given multipliers instead
        (what the parity tests do: the reference draws them from the unseeded global ``np.random``); ``charger_detail``
        also keeps every charger's electricity consumption and requested energy of the step (``charger_out``).

        ``env_pitch`` (`cl_dims.env_pitch`): floats between consecutive building rows of the state / output planes.  Default: n_env -- except for
        battery + PV districts whose batch is a large power-of-two multiple (n_env a multiple of 65 536, from 524 288 envs up), where the rows
        are padded by 256 envs so that their byte stride is not a multiple of 256 KiB: the step's 153 streams alias in the memory system at
        such strides (17 x 524 288: 74.2 -> 63.7 us, 17 x 1 048 576: 124.0 -> 121.5 us; at 262 144 envs, where the Infinity Cache still holds
        most of the step, the pad costs 3 % and is not applied: profiles/r05c_*).  `state` / `out_bldg` stay `[planes, n_bldg, n_env]` tensors (views of the padded storage).

        ``f64_maps`` (`CLD_F64_MAPS`): evaluate the battery map in float64 with float32 rounding where the reference's float32 series
        round -- the reference's own precision model (energy_model.py:1027-1141), for free-running parity at 1e-4; slower launches,
        no fused rollout kernel, no flexible loads.  ``f64_maps='chain'`` (`CLD_F64_CHAIN`): only the battery's soc chain in float64
        and the degraded capacity carried as the capacity LOSS in its float32 plane (`degraded_capacity` converts) -- not bit-identical
        but inside 1e-4 free-running on every fixture, the default three state planes, every step kernel and the fused rollout.
        **Default (None, round 6): ``'chain'`` wherever the district admits it** (`chain_supported`: no EV chargers / washing machines), the
        all-fp32 map otherwise.  The fp32 map (``f64_maps=False``) is 1.16 x faster per step at the headline shape and holds 1e-4 teacher-forced,
        but free-running it drifts to 6.9 x the bar on `net` over the 8 759-step year of BASELINE config 1 (tests/test_gpu_parity.py::
        test_full_year_free_running_every_step): a throughput mode to opt into, not the default.

        ``check`` (`CLD_CHECK`, a debug mode): evaluate the reference's runtime assertions inside the step (flexibility >= 0 under an outage,
        device-consumption polarity, non-negative non-shiftable load: building.py:665, 1831-1835; energy_model.py:146-148) and keep one word of
        `abi.CLV_*` bits per unit (:attr:`violations`).  Needs ``detail=True`` and at most 32 buildings; runs the general step kernel."""
        self.lib = _lib.load()                      # raises if the HIP extension is not built
        if not torch.cuda.is_available():
            raise _lib.EngineUnavailable('no HIP device visible: the step engine only runs on the GPU')
        if n_env % 4:
            raise ValueError('n_env must be a multiple of 4 (pad the batch)')
        if env_offset < 0 or env_offset + n_env > 2 ** 32:
            raise ValueError(f'env_offset={env_offset} with n_env={n_env} leaves the 32-bit env index of the random streams')
        self.device = torch.device(device)
        self.n_env = int(n_env)
        self.n_bldg = int(tables.params.shape[0])
        self.n_ts_rows = int(tables.ts.shape[0])
        self.n_steps = self.n_ts_rows if n_steps is None else int(n_steps)
        if not 0 < self.n_steps <= self.n_ts_rows:
            raise ValueError(f'n_steps={self.n_steps} outside (0, {self.n_ts_rows}]')
        self.env_row0 = None
        if env_row0 is not None:
            row0 = np.asarray(env_row0, dtype=np.int64).reshape(-1)
            n_blocks = -(-self.n_env // abi.CL_ROW0_BLOCK)
            if row0.shape[0] != n_blocks:
                raise ValueError(f'env_row0 needs {n_blocks} entries (one per {abi.CL_ROW0_BLOCK} envs), got {row0.shape[0]}')
            if row0.min() < 0 or row0.max() + self.n_steps > self.n_ts_rows:
                raise ValueError(f'env_row0 + n_steps must stay inside the {self.n_ts_rows} table rows')
            self.env_row0_host = row0.astype(np.int32)
        self.flex_tables = tables.flex
        if reward == 'Electric_Vehicles_Reward_Function' and self.flex_tables is None:
            raise ValueError('Electric_Vehicles_Reward_Function needs a district with EV chargers')
        if n_act_cols is None:
            cols = tables.params.view(np.int32)[:, abi.CLP_ACT_COOL_STO:abi.CLP_ACT_COH_DEV + 1]
            n_act_cols = int(cols.max()) + 1 if self.flex_tables is None else self.flex_tables.n_act_cols
        self.n_act_cols = n_act_cols
        self.reward = reward
        flags = (REWARD_KINDS[reward] << abi.CLD_REWARD_SHIFT)
        flags |= abi.CLD_REF_T0_QUIRK if t0_quirk else 0
        flags |= abi.CLD_KPI if kpi else 0
        flags |= abi.CLD_CENTRAL_AGENT if central_agent else 0      # only read by the CLR_EV reward
        if f64_maps is None:
            f64_maps = 'chain' if self.chain_supported(tables) else False
        if f64_maps not in (False, True, 'ref', 'chain'):
            raise ValueError("f64_maps must be None (default: 'chain' where supported), False, True / 'ref' (CLD_F64_MAPS) or 'chain' (CLD_F64_CHAIN)")
        self.f64_chain = f64_maps == 'chain'
        self.f64_maps = bool(f64_maps) and not self.f64_chain
        flags |= abi.CLD_F64_MAPS if self.f64_maps else 0
        flags |= abi.CLD_F64_CHAIN if self.f64_chain else 0
        if f64_maps and tables.flex is not None:
            raise NotImplementedError('f64_maps is not implemented for districts with EV chargers / washing machines')
        if self.f64_chain:
            has_batt = (tables.params[:, abi.CLP_FLAGS] & abi.CLF_BATTERY) != 0
            valid = tables.params[:, abi.CLP_C_FIRST:abi.CLP_C_LAST + 1].copy().view(np.float64)[:, abi.CLPC_VALID]
            if np.any(has_batt & (valid != 1.0)):
                raise NotImplementedError("f64_maps='chain' needs battery curves with ascending breakpoints that end at 1 and power fractions <= 1 "
                                          "(the ramp form of the float64 chain); use f64_maps=True for this district")
        self.kpi = kpi
        bflags = tables.params[:, abi.CLP_FLAGS]
        heavy = abi.CLF_THERMAL | abi.CLF_OUTAGE | abi.CLF_DYNAMICS
        self.lean = not bool(np.any(bflags & heavy)) and not bool(np.any(tables.ts[:, :, [abi.CLT_COOL_DEM, abi.CLT_HEAT_DEM, abi.CLT_DHW_DEM]]))
        flags |= abi.CLD_LEAN if self.lean else 0
        # the streaming KPI passes read the detail planes -- except for battery + PV districts of up to 32 buildings, whose step kernel
        # updates the per-building accumulators itself (cl_step_lean_kpi_kernel): no detail planes, no second pass
        kpi_in_step = kpi and self.lean and self.n_bldg <= 32 and self.flex_tables is None and not self.f64_maps     # (fp32 or the float64 chain)
        # ... and for thermal / outage districts stepped by the one-env-per-lane thermal kernel (cl_step_full_kpi_kernel; any launch override
        # that selects another kernel falls back to the detail subset + the KPI launch)
        kpi_in_full_step = (kpi and not self.lean and self.n_bldg <= 32 and self.flex_tables is None and not f64_maps
                            and not any((tuning or {}).get(k) for k in ('vec', 'kpi_passes', 'no_chunks')) and (tuning or {}).get('full_variant') != 1)
        # `detail`: True = every detail plane (observations, evaluate()'s series, parity tests, reward plugins); 'min' = only the planes
        # another kernel of the path reads (CLD_DETAIL_MIN: baseline / expected / served for the KPI pass, delivered demands for the LSTM
        # stage) -- which is also what streaming KPIs alone ask for
        if detail not in (True, False, 'min'):
            raise ValueError("detail must be True, False or 'min'")
        if detail is False and kpi and not kpi_in_step and not kpi_in_full_step:
            detail = 'min'
        flags |= abi.CLD_WRITE_DETAIL if detail else 0
        flags |= abi.CLD_DETAIL_MIN if detail == 'min' else 0
        self.check = bool(check)
        if self.check:
            if detail is not True or self.n_bldg > 32:
                raise ValueError('check=True (CLD_CHECK) needs detail=True and a district of at most 32 buildings (the violation words use the scratch plane '
                                 'of building-chunked launches)')
            flags |= abi.CLD_CHECK
            if kpi:
                tuning = {**(tuning or {}), 'kpi_passes': (tuning or {}).get('kpi_passes') or 1}     # the KPI launch after the (general) step kernel
        # ... and keeps the env-independent sums of such a district (baseline, expected energy, baseline district series) once per block
        # of CL_ROW0_BLOCK envs, at the block's first env (include/citylearn_amd.h, CLD_KPI): `kpi.finalize_streaming(shared_baseline=True)`
        self.kpi_shared_baseline = bool(kpi_in_step and not detail)
        self.detail = detail
        es_cols = tables.params.view(np.int32)[:, abi.CLP_ACT_ELEC_STO]
        if np.array_equal(es_cols, np.arange(self.n_bldg)):          # one battery action per building, building order
            flags |= abi.CLD_ES_COL_IS_BLDG
        with torch.cuda.device(self.device):
            if env_row0 is not None:
                self.env_row0 = torch.from_numpy(self.env_row0_host).to(self.device)
        # launch-geometry overrides of tests / tuning scripts travel with every call (`cl_dims.tuning`); all zero = defaults
        self.tuning = _lib.Tuning()
        for key, value in (tuning or {}).items():
            if key not in dict(_lib.Tuning._fields_) or key == 'kernel_name':
                raise ValueError(f'unknown tuning field {key!r}')
            setattr(self.tuning, key, int(value))
        # row pitch of the state / output planes (cl_dims.env_pitch): only where the library implements one
        pitch_ok = self.lean and not kpi and self.flex_tables is None and not self.f64_maps and not detail and self.n_bldg <= 32
        if env_pitch is None:
            env_pitch = self.n_env + 256 if (pitch_ok and self.n_env >= 524288 and self.n_env % 65536 == 0) else self.n_env
        env_pitch = int(env_pitch)
        if env_pitch != self.n_env and (not pitch_ok or env_pitch < self.n_env or env_pitch % 4):
            raise ValueError(f'env_pitch={env_pitch}: a multiple of 4 >= n_env, for battery + PV districts of up to 32 buildings without detail planes, '
                             'streaming KPIs, flexible loads or f64_maps=True')
        self.env_pitch = env_pitch
        self.dims = _lib.Dims(self.n_env, self.n_bldg, self.n_steps, self.n_act_cols, flags, self.n_ts_rows,
                              None if self.env_row0 is None else self.env_row0.data_ptr(), ctypes.pointer(self.tuning), int(env_offset),
                              0 if env_pitch == self.n_env else env_pitch, 0)
        with torch.cuda.device(self.device):
            self.params = torch.from_numpy(tables.params.view(np.int32).copy()).to(self.device)
            self.ts = torch.from_numpy(np.ascontiguousarray(tables.ts)).to(self.device)
            self._state_store = torch.zeros((abi.CL_NS, self.n_bldg, self.env_pitch), dtype=torch.float32, device=self.device)
            self._out_store = torch.zeros((abi.CL_NO, self.n_bldg, self.env_pitch), dtype=torch.float32, device=self.device)
            # (views of the padded storage when the rows carry a pitch: same base address, row stride env_pitch)
            self.state = self._state_store[:, :, :self.n_env]
            self.out_bldg = self._out_store[:, :, :self.n_env]
            self._out_env = torch.zeros((abi.CL_NQ, self.n_env), dtype=torch.float32, device=self.device)
            self.kpi_bldg = torch.zeros((abi.CL_NKB, self.n_bldg, self.n_env), dtype=torch.float32, device=self.device) if kpi else None
            self.kpi_env = torch.zeros((abi.CL_NKE, self.n_env), dtype=torch.float32, device=self.device) if kpi else None
            self.flex = None
            if self.flex_tables is not None:
                self._init_flex(ev_reward_weights, ev_drift, ev_seed, charger_detail, ev_penalty_coefficient)
        self._device_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or (lambda i: torch.cuda.current_stream(i).cuda_stream)
        # per-step call arguments that never change: computed once (ctypes conversions dominate an 8 us kernel otherwise)
        self._step_head = (ctypes.byref(self.dims), _ptr(self.params), _ptr(self.ts), _ptr(self.state))
        self._step_tail = (_ptr(self.out_bldg), _ptr(self._out_env), _ptr(self.kpi_bldg), _ptr(self.kpi_env))
        # deferred finish (`tuning={'finish': 3}`, districts of more than 32 buildings): `step` leaves the district sums of its step to the
        # next launch; reading `out_env` (or any view of it) folds the pending one first (`finish`).  `_deferred` is read from the tuning
        # block at every call: the library does the same (cl_dims.tuning travels with the call), so the two cannot disagree.
        self._pending_t = None
        self._flex_ref = None if self.flex is None else ctypes.byref(self.flex)
        self.act_low = self.act_high = None         # bounds of the on-device rollout policy (set_action_limits)
        self._policy_actions = None                 # scratch planes of cl_rollout_seq_f32 (four steps of policy draws)
        self.t = 0
        self.reset()

    def _init_flex(self, weights, drift, seed: int, charger_detail: bool, penalty_coefficient: float):
        """Device copies of the flexible-load tables + their state planes (`cl_flex`, include/citylearn_amd.h)."""
        from .flex import reward_weights
        ft = self.flex_tables
        if ft.n_rows < self.n_ts_rows:
            raise ValueError('flexible-load tables are shorter than the step tables')
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        n_ev, n_fb = len(ft.ev_names), ft.flex_bldg.shape[0]
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.device)
        self._flex_buffers = dict(
            ev_params=dev(ft.ev_params.view(np.int32)), ev_ts=dev(ft.ev_ts), charger_params=dev(ft.charger_params.view(np.int32)),
            charger_ts=dev(ft.charger_ts), wm_params=dev(ft.wm_params.view(np.int32)), wm_ts=dev(ft.wm_ts),
            cons_params=None if ft.cons_params is None else dev(ft.cons_params.view(np.int32)),
            ev_state=z(abi.CL_NEVS, max(n_ev, 1), self.n_env), wm_state=z(n_fb * abi.CL_MAXW, self.n_env),
            flex_out=z(abi.CL_NX, n_fb, self.n_env),
            charger_out=z(2, n_fb * abi.CL_MAXC, self.n_env) if charger_detail else None)
        self._charger_slot = torch.from_numpy(ft.charger_slot.astype(np.int64)).to(self.device)
        self.ev_drift = None
        if drift is not None:
            drift = np.asarray(drift, dtype=np.float32)
            if drift.shape != (ft.n_rows, n_ev):
                raise ValueError(f'ev_drift shape {drift.shape} != {(ft.n_rows, n_ev)}')
            self.ev_drift = dev(drift)
        b = self._flex_buffers
        self.flex = _lib.Flex(
            n_ev, n_fb, ft.n_rows, 0, _ptr(b['ev_params']), _ptr(b['ev_ts']), _ptr(b['charger_params']),
            _ptr(b['charger_ts']), _ptr(b['wm_params']), _ptr(b['wm_ts']), _ptr(b['cons_params']), _ptr(b['ev_state']),
            _ptr(b['wm_state']), _ptr(b['flex_out']), _ptr(b['charger_out']), _ptr(self.ev_drift), int(seed) & (2 ** 64 - 1),
            (ctypes.c_float * 8)(*reward_weights(weights, penalty_coefficient).tolist()))
        self.ev_state, self.wm_state = b['ev_state'], b['wm_state']
        self.flex_out = b['flex_out']

    @property
    def charger_out(self) -> torch.Tensor:
        """``[2, n_charger, n_env]``: electricity consumption and requested energy of every charger at the last step
        (chargers in building order; construct with ``charger_detail=True``)."""
        raw = self._flex_buffers['charger_out']
        if raw is None:
            raise RuntimeError('construct StepEngine(..., charger_detail=True) to keep per-charger outputs')
        return raw[:, self._charger_slot]

    # ---------------------------------------------------------------------------------------------------
    def trace_kernels(self, on: bool = True):
        """Diagnostics (`cl_tuning.kernel_name`): have every step / rollout / LSTM call of this engine report which kernel
        instantiation(s) it launched; read them back with :attr:`last_kernels`.  What `bench.py` prints as `roofline.kernel`."""
        self._kernel_name = ctypes.create_string_buffer(abi.CL_KERNEL_NAME_LEN) if on else None
        self.tuning.kernel_name = ctypes.cast(self._kernel_name, ctypes.c_void_p) if on else None

    @property
    def last_kernels(self) -> str:
        """'+'-separated kernel instantiations of the last call, in rocprofv3's spelling (after `trace_kernels()`)."""
        if getattr(self, '_kernel_name', None) is None:
            raise RuntimeError('call trace_kernels() first')
        return self._kernel_name.value.decode()

    def _stream(self) -> int:
        # raw handle of torch's current stream on this device (what torch.cuda.current_stream(...).cuda_stream returns, without
        # building the Stream object: this runs once per env step)
        return self._raw_stream(self._device_index)

    def _on_device(self):
        """Context that makes the engine's device current -- a no-op object when it already is (the common case)."""
        return _NULL_CONTEXT if torch.cuda.current_device() == self._device_index else torch.cuda.device(self.device)

    def reset(self):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cl_reset_f32(ctypes.byref(self.dims), _ptr(self.params), _ptr(self.state), _ptr(self.kpi_bldg), _ptr(self.kpi_env),
                                             self._stream()))
            if self.flex is not None:
                _lib.check(self.lib.cl_flex_reset_f32(ctypes.byref(self.dims), ctypes.byref(self.flex), self._stream()))
                self.flex_out.zero_()
            # the output planes are what a 'planes' observation hands out: an episode must not start on the previous one's last step
            self.out_bldg.zero_()
            self._out_env.zero_()
        self._pending_t = None
        self.t = 0

    # ---- checkpoint / restore ("torch.save of the tensor dict is a complete checkpoint") -------------------------
    _CHECKPOINT_FORMAT = 1

    def _signature(self) -> dict:
        """What has to agree between the engine a checkpoint was taken from and the one it is loaded into."""
        return {'n_env': self.n_env, 'n_bldg': self.n_bldg, 'n_steps': self.n_steps, 'flags': int(self.dims.flags), 'n_act_cols': self.n_act_cols,
                'env_offset': int(self.dims.env_offset), 'flex': self.flex is not None, 'kpi': bool(self.kpi),
                'params_crc': int(self.params.to(torch.int64).sum().item()) & 0xFFFFFFFF}

    def state_dict(self) -> dict:
        """Everything the device carries from one step to the next, as (cloned) tensors + a few host scalars: the state planes, the output
        planes of the last step (what the next observation and `evaluate()` read; a pending deferred fold is finished first), the district
        sums, the streaming KPI accumulators, the flexible-load state (EV batteries, washing-machine progress, the drift seed) and the step
        counter.  The random streams (rollout policy, EV drift) are counter-based -- Philox keyed by (seed, env, column, step) -- so they have
        no state of their own: a restored engine draws the same numbers.  `torch.save(engine.state_dict(), path)` is a complete checkpoint
        (the reference's users pickle the whole env: citylearn/__main__.py:291-299)."""
        self.finish()
        c = lambda x: None if x is None else x.detach().clone()
        sd = {'format': self._CHECKPOINT_FORMAT, 'signature': self._signature(), 't': int(self.t),
              'state': c(self._state_store), 'out_bldg': c(self._out_store), 'out_env': c(self._out_env),
              'kpi_bldg': c(self.kpi_bldg), 'kpi_env': c(self.kpi_env)}
        if self.flex is not None:
            b = self._flex_buffers
            sd['flex'] = {'ev_state': c(b['ev_state']), 'wm_state': c(b['wm_state']), 'flex_out': c(b['flex_out']),
                          'charger_out': c(b['charger_out']), 'seed': int(self.flex.seed)}
        return sd

    def load_state_dict(self, sd: Mapping) -> None:
        """Restore :meth:`state_dict` into THIS engine's buffers (same district, batch size and flags: checked), in place -- captured
        hipGraphs that point at them stay valid.  The next `step` continues bit-identically to the engine the checkpoint was taken from."""
        if sd.get('format') != self._CHECKPOINT_FORMAT:
            raise ValueError(f"checkpoint format {sd.get('format')!r}, this build reads {self._CHECKPOINT_FORMAT}")
        mine = self._signature()
        diff = {k: (v, mine.get(k)) for k, v in dict(sd['signature']).items() if mine.get(k) != v}
        if diff:
            raise ValueError(f'checkpoint does not belong to this engine (saved, here): {diff}')
        with torch.cuda.device(self.device):
            for key, dst in (('state', self._state_store), ('out_bldg', self._out_store), ('out_env', self._out_env), ('kpi_bldg', self.kpi_bldg),
                             ('kpi_env', self.kpi_env)):
                src = sd.get(key)
                if (src is None) != (dst is None) or (src is not None and tuple(src.shape) != tuple(dst.shape)):
                    raise ValueError(f'checkpoint tensor {key!r} does not fit this engine')
                if dst is not None:
                    dst.copy_(src.to(self.device))
            if self.flex is not None:
                f, b = sd['flex'], self._flex_buffers
                for key in ('ev_state', 'wm_state', 'flex_out', 'charger_out'):
                    if b[key] is not None and f.get(key) is not None:
                        b[key].copy_(f[key].to(self.device))
                self.flex.seed = int(f['seed'])
        self._pending_t = None
        self.t = int(sd['t'])

    @staticmethod
    def chain_supported(tables: EpisodeTables) -> bool:
        """Whether `f64_maps='chain'` (CLD_F64_CHAIN) can step this district: no EV chargers / washing machines, battery curves in the shape
        the chain's ramp form assumes (every shipped dataset's)."""
        if tables.flex is not None:
            return False
        has_batt = (tables.params[:, abi.CLP_FLAGS] & abi.CLF_BATTERY) != 0
        valid = tables.params[:, abi.CLP_C_FIRST:abi.CLP_C_LAST + 1].copy().view(np.float64)[:, abi.CLPC_VALID]
        return not bool(np.any(has_batt & (valid != 1.0)))

    @property
    def _deferred(self) -> bool:
        return int(self.tuning.finish) == 3 and self.n_bldg > 32

    def step(self, actions: torch.Tensor, t: Optional[int] = None):
        """Advance every (env, building) by one step.  ``actions``: float32 ``[n_act_cols, n_env]`` on the
        engine's device (any 2-D strides; ``[n_env, n_act_cols].T`` works too)."""
        t = self.t if t is None else t
        if actions.dtype != torch.float32 or actions.device != self.device:
            raise TypeError('actions must be a float32 tensor on the engine device')
        if tuple(actions.shape) != (self.n_act_cols, self.n_env):
            raise ValueError(f'actions shape {tuple(actions.shape)} != {(self.n_act_cols, self.n_env)}')
        sc, se = actions.stride()
        with self._on_device():
            if self._flex_ref is not None:
                rc = self.lib.cl_step_flex_f32(*self._step_head, actions.data_ptr(), sc, se, *self._step_tail, self._flex_ref,
                                               int(t), self._stream())
            else:
                rc = self.lib.cl_step_f32(*self._step_head, actions.data_ptr(), sc, se, *self._step_tail, int(t), self._stream())
        if rc:
            _lib.check(rc)
        if self._deferred:
            self._pending_t = int(t)
        self.t = t + 1

    def finish(self):
        """Deferred finish (`tuning={'finish': 3}`): fold the chunk partial sums the last `step` left behind into `out_env`
        (`cl_finish_f32`; one small launch on the current stream).  Called by every read of `out_env` / `district_*`; call it yourself
        at the end of a step sequence captured into a hipGraph (a replay runs no Python).  Nothing to do in any other mode."""
        if self._pending_t is None:
            return
        with self._on_device():
            _lib.check(self.lib.cl_finish_f32(ctypes.byref(self.dims), _ptr(self.out_bldg), _ptr(self._out_env), self._pending_t, self._stream()))
        self._pending_t = None

    @property
    def out_env(self) -> torch.Tensor:
        """``[CL_NQ, n_env]`` district sums of the last step (after a deferred step: folded first, see `finish`)."""
        self.finish()
        return self._out_env

    def step_observe(self, actions: torch.Tensor, writer, t: Optional[int] = None) -> torch.Tensor:
        """`step` followed by ``writer.write(t + 1)`` for an `ObservationWriter` over the COMPACT tables (every column env-dependent):
        one C call, `cl_step_observe_f32` -- and one launch where the step runs as a single lean launch at four envs per lane."""
        t = self.t if t is None else t
        if self._flex_ref is not None or writer.stage is not None or writer.n_deps != writer.n_cols or writer.n_cols == 0:
            self.step(actions, t)
            return writer.write(min(t + 1, self.n_steps - 1))
        if actions.dtype != torch.float32 or actions.device != self.device:
            raise TypeError('actions must be a float32 tensor on the engine device')
        if tuple(actions.shape) != (self.n_act_cols, self.n_env):
            raise ValueError(f'actions shape {tuple(actions.shape)} != {(self.n_act_cols, self.n_env)}')
        sc, se = actions.stride()
        row = min(t + 1, self.n_steps - 1)
        with self._on_device():
            rc = self.lib.cl_step_observe_f32(*self._step_head, actions.data_ptr(), sc, se, *self._step_tail, int(t),
                                              writer.table.data_ptr(), writer.col_src.data_ptr(), writer.col_scale.data_ptr(),
                                              ctypes.cast(writer._deps, ctypes.c_void_p), writer.n_deps, writer._buffer.data_ptr(), writer.n_cols,
                                              writer.pitch, writer.n_rows, int(row), self._stream())
        if rc:
            _lib.check(rc)
        if self._deferred:
            self._pending_t = int(t)        # (cl_step_observe_f32 runs the same step launch: a chunked district defers its sums here too)
        self.t = t + 1
        return writer.obs

    def set_action_limits(self, low, high):
        """Bounds of the on-device uniform random policy of :meth:`rollout` (``[n_act_cols]`` each)."""
        self.act_low = torch.as_tensor(np.asarray(low, dtype=np.float32)).to(self.device).contiguous()
        self.act_high = torch.as_tensor(np.asarray(high, dtype=np.float32)).to(self.device).contiguous()
        assert self.act_low.numel() == self.n_act_cols == self.act_high.numel()

    def rollout(self, k_steps: int, actions: Optional[torch.Tensor] = None, seed: int = 0,
                ret_env: Optional[torch.Tensor] = None, t0: Optional[int] = None, fused: Optional[bool] = None):
        """Fused K-step rollout in one launch (`cl_rollout_f32`): state stays in registers between steps.  Districts of more than
        32 battery + PV / 16 thermal buildings run it building-chunked (one more tiny launch per K steps folds the chunks' district
        sums and returns).  Districts with flexible loads, streaming KPIs (``kpi=True``), the float64 battery map, or a reward that
        couples the buildings of a chunked district inside a step (MARL) run the same K steps as K x (policy, [flex], step, [kpi])
        launches (`cl_rollout_seq_f32`), same action streams; ``fused=False`` asks for that sequence explicitly.

        ``actions``: open-loop float32 tensor ``[k_steps, n_act_cols, n_env]`` (any strides), or ``None`` for the
        on-device policy ``a = low + u (high - low)``, ``u = Philox4x32-10(seed; env, column, t)``.
        ``ret_env`` (``[n_env]``, optional) accumulates the district reward summed over the K steps."""
        t0 = self.t if t0 is None else t0
        st = (0, 0, 0)
        if actions is not None:
            if actions.dtype != torch.float32 or actions.device != self.device:
                raise TypeError('actions must be a float32 tensor on the engine device')
            if tuple(actions.shape) != (k_steps, self.n_act_cols, self.n_env):
                raise ValueError(f'actions shape {tuple(actions.shape)} != {(k_steps, self.n_act_cols, self.n_env)}')
            st = actions.stride()
        elif self.act_low is None:
            raise ValueError('call set_action_limits(low, high) before using the on-device policy')
        full = not self.lean or bool(self.dims.flags & abi.CLD_WRITE_DETAIL)
        chunked = self.n_bldg > (16 if full else 32)
        if fused is None:
            fused = not (self.flex is not None or self.kpi or self.f64_maps or (chunked and self.reward == 'MARL'))
        if not fused:
            if actions is None and self._policy_actions is None:
                self._policy_actions = torch.empty((4, self.n_act_cols, self.n_env), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.cl_rollout_seq_f32(
                    ctypes.byref(self.dims), _ptr(self.params), _ptr(self.ts), _ptr(self.state), _ptr(actions), st[0], st[1], st[2],
                    _ptr(self.act_low), _ptr(self.act_high), int(seed) & (2 ** 64 - 1),
                    _ptr(None if actions is not None else self._policy_actions), _ptr(self.out_bldg), _ptr(self._out_env), _ptr(ret_env),
                    _ptr(self.kpi_bldg), _ptr(self.kpi_env), self._flex_ref, int(t0), int(k_steps), self._stream()))
            self.t = t0 + k_steps
            self._pending_t = None          # (cl_rollout_seq_f32 finishes its last step itself)
            return
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cl_rollout_f32(
                ctypes.byref(self.dims), _ptr(self.params), _ptr(self.ts), _ptr(self.state), _ptr(actions), st[0], st[1], st[2],
                _ptr(self.act_low), _ptr(self.act_high), int(seed) & (2 ** 64 - 1),
                _ptr(self.out_bldg), _ptr(self._out_env), _ptr(ret_env), int(t0), int(k_steps), self._stream()))
        self._pending_t = None              # (a chunked fused rollout folds its last step's district sums itself)
        self.t = t0 + k_steps

    def step_many(self, actions: torch.Tensor, t0: Optional[int] = None):
        """``actions.shape[0]`` consecutive env steps enqueued by ONE C call (`cl_rollout_seq_f32` with an open-loop action tensor
        ``[k, n_act_cols, n_env]``, any strides): the same launches as k calls of :meth:`step`, without k trips through Python and ctypes
        -- for callers that already hold the next k actions (replay, evaluation of a fixed schedule, `bench.py`'s short timed regions)."""
        t0 = self.t if t0 is None else t0
        k = int(actions.shape[0])
        if actions.dtype != torch.float32 or actions.device != self.device or tuple(actions.shape[1:]) != (self.n_act_cols, self.n_env):
            raise ValueError(f'actions must be float32 [k, {self.n_act_cols}, {self.n_env}] on {self.device}')
        st = actions.stride()
        with self._on_device():
            rc = self.lib.cl_rollout_seq_f32(ctypes.byref(self.dims), _ptr(self.params), _ptr(self.ts), _ptr(self.state), actions.data_ptr(), st[0], st[1], st[2],
                                             None, None, 0, None, _ptr(self.out_bldg), _ptr(self._out_env), None, _ptr(self.kpi_bldg), _ptr(self.kpi_env),
                                             self._flex_ref, int(t0), k, self._stream())
        if rc:
            _lib.check(rc)
        self._pending_t = None              # (cl_rollout_seq_f32 finishes its last step itself)
        self.t = t0 + k

    # convenient views ------------------------------------------------------------------------------------
    @property
    def soc(self) -> torch.Tensor:
        return self.state[abi.CLS_B_SOC]

    @property
    def degraded_capacity(self) -> torch.Tensor:
        """``[n_bldg, n_env]`` Battery.degraded_capacity [kWh].  Under `f64_maps='chain'` the state plane carries the capacity LOSS
        (capacity - degraded capacity: what lets a float32 plane hold the reference's float64 attribute): converted here."""
        plane = self.state[abi.CLS_B_DEGCAP]
        if not self.f64_chain:
            return plane
        cap = self.params[:, abi.CLP_L_CAP].view(torch.float32)
        return cap[:, None] - plane

    @property
    def violations(self) -> torch.Tensor:
        """``[n_bldg, n_env]`` int32 words of `abi.CLV_*` bits: the reference assertions the last step tripped (``check=True``; 0 = none)."""
        if not self.check:
            raise RuntimeError('construct StepEngine(..., check=True, detail=True) to evaluate the reference assertions (CLD_CHECK)')
        return self.out_bldg[abi.CLO_RESERVED].view(torch.int32)

    @property
    def net(self) -> torch.Tensor:
        return self.out_bldg[abi.CLO_NET]

    @property
    def reward_bldg(self) -> torch.Tensor:
        return self.out_bldg[abi.CLO_REWARD]

    @property
    def district_net(self) -> torch.Tensor:
        return self.out_env[abi.CLQ_NET]

    @property
    def district_reward(self) -> torch.Tensor:
        return self.out_env[abi.CLQ_REWARD]

    def algorithmic_bytes_per_unit(self) -> float:
        """HBM bytes one (env, building) unit must move in one `cl_step_f32` launch (mode A-min):
        state planes read+written, action columns read, net + reward written, district sums written."""
        flags = self.params[:, abi.CLP_FLAGS].cpu().numpy().view(np.uint32)
        planes = 0.0
        for f in flags:
            planes += (5 if self.f64_maps else 3) * bool(f & abi.CLF_BATTERY) + bool(f & abi.CLF_COOL_STO) + bool(f & abi.CLF_HEAT_STO) + bool(f & abi.CLF_DHW_STO)
        planes /= self.n_bldg
        step = 8.0 * planes + 4.0 * self.n_act_cols / self.n_bldg + 8.0 + 4.0 * abi.CL_NQ / self.n_bldg
        if not self.kpi:
            return step
        # mode A-kpi.  Per unit, read + written every step: the four control sums (positive net, net, emission, cost) and, where
        # the baseline depends on the env (thermal / outage districts), its four sums and the unserved / expected energy sums: 4 or 10
        # accumulators.  Per env and district series (control; + baseline where it depends on the env): the seven values that move every
        # step -- previous value, ramping sum, open day's sum and maximum, open month's sum and maximum, all-time peak (the closed-group
        # sums move once per 24 / 730 steps and are not counted).
        moving = 7.0
        if self.kpi_shared_baseline:
            return step + 32.0 + 8.0 * moving / self.n_bldg
        return step + 80.0 + 16.0 * moving / self.n_bldg
