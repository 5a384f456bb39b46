"""`VectorCityLearnEnv`: E independent CityLearn districts stepped together on one GPU, tensors in / tensors out.

This is the form the engine is designed for (one `cl_step_f32` launch advances every (env, building) unit); the
list-based `CityLearnEnv` is the single-district compatibility surface.  All envs replay the same episode window
of the schema; they differ by their actions (and therefore storage trajectories).
"""
from __future__ import annotations

from typing import Any, Dict, Mapping, Optional, Tuple, Union

import numpy as np
import torch

from . import abi
from .engine import REWARD_KINDS, StepEngine
from .reward_function import resolve as resolve_reward
from .schema import DistrictSpec, load_district


class VectorCityLearnEnv:
    """Batched environment.

    * ``actions``: float32 tensor ``[n_act_cols, n_envs]`` (coalesced layout) or ``[n_envs, n_act_cols]`` (policy
      layout, read through strides); columns follow the reference's central-agent order (building-major).
    * ``step`` returns ``(obs, reward, terminated, truncated, info)`` where ``reward`` is ``[n_envs]`` (central agent:
      sum over buildings) or ``[n_bldg, n_envs]``; ``obs`` is a dict of device tensors: ``'exogenous'`` ``[n_exo]`` -- the
      env-independent observation values of the next time step shared by all envs -- and the env-dependent planes
      ``'electrical_storage_soc'`` ``[n_bldg, n_envs]``, ``'net_electricity_consumption'`` ``[n_bldg, n_envs]``, tank SoCs.
    """

    def __init__(self, schema: Union[str, Mapping[str, Any], DistrictSpec], n_envs: int, device: str = 'cuda:0',
                 reference_quirks: bool = True, kpi: bool = False, observations: str = 'planes',
                 normalize_observations: bool = False, observation_mode: str = 'current',
                 env_episode_offsets=None, ev_seed: Optional[int] = None, ev_soc_drift=None, env_offset: int = 0, f64_maps=None,
                 **kwargs: Any):
        """`observations`: ``'planes'`` (default) returns the dict of device tensors described above without
        materialising anything; ``'tensor'`` returns the Gym observation tensor ``[n_envs, n_obs]`` written by
        `cl_observe_f32` -- columns = `observation_names` (the reference's central-agent order when
        ``central_agent``, else the agents' vectors concatenated), optionally min-max / sin-cos normalised like
        `NormalizedObservationWrapper` (`normalize_observations`); ``'compact'`` returns the same observation factorised --
        ``{'shared': [n_obs] row of the step (one per env block with episode offsets), 'dependent': [n_envs, n_dep] values of the
        env-dependent columns, 'columns': their indices}`` -- which moves ~7 % of the bytes of the full matrix (17 x 65 536:
        9 MB instead of 125 MB per step); `materialize(obs)` expands it to the ``'tensor'`` form.  `observation_mode`: ``'current'`` pairs the
        exogenous values of step t+1 with the SoC / net just computed; ``'reference'`` reproduces the reference's
        stale read of the t+1 slots.
        `env_episode_offsets`: ``None`` -- every env replays the same episode window (the reference's sequential episodes);
        ``'rolling'`` / ``'random'`` / an int array with one entry per block of ``abi.CL_ROW0_BLOCK`` envs -- blocks replay
        DIFFERENT windows of ``episode_time_steps`` rows of the simulation period at once (start rows relative to
        ``simulation_start_time_step``; ``'random'`` redraws them at every `reset`).
        Districts with EV chargers / washing machines run the extra `cl_flex_kernel` launch per step;
        `ev_seed` keys the per-(env, EV, step) N(1, 0.2) drift of unconnected EVs (default: the schema's random_seed, advanced
        per episode), `ev_soc_drift` ([table rows, n_ev]) replays given multipliers for every env instead.
        `env_offset`: index of this shard's first env in a multi-GPU batch (`parallel.shard_envs(total, rank, world)[0]`): random streams
        (rollout policy, EV drift) are keyed by it + the local env index, so shards with one seed draw disjoint streams.
        `f64_maps`: precision model of the battery map.  Default (None): ``'chain'`` (`CLD_F64_CHAIN`) wherever the district
        admits it -- the mode that holds 1e-4 free-running over whole episodes, the full year of BASELINE config 1 included -- else the fp32
        map; ``False``: the all-fp32 map (1.16 x faster per step, drifts past 1e-4 free-running: a throughput mode); ``True`` (`CLD_F64_MAPS`):
        the reference's own mixed float64 / float32 precision, battery state bit-identical over a free-running episode at about 3 x the step time."""
        if observations not in ('planes', 'tensor', 'compact'):
            raise ValueError("observations must be 'planes', 'tensor' or 'compact'")
        self._compact = observations == 'compact'
        self.spec = schema if isinstance(schema, DistrictSpec) else load_district(schema, **kwargs)
        self.n_envs = int(n_envs)
        self.device = torch.device(device)
        self.reference_quirks = reference_quirks
        self.kpi = kpi
        self.central_agent = self.spec.central_agent
        self.env_episode_offsets = env_episode_offsets
        self._ev_seed, self._ev_drift = ev_seed, ev_soc_drift
        self.f64_maps = f64_maps if f64_maps in ('chain', 'ref', None) else bool(f64_maps)      # (None: 'chain' where supported -- resolved by StepEngine)
        self.env_offset = int(env_offset)      # first env of this shard in the whole batch (multi-GPU: parallel.shard_envs(...)[0])
        if self.env_offset < 0 or self.env_offset + self.n_envs > 2 ** 32:
            raise ValueError(f'env_offset={env_offset} with n_envs={n_envs} leaves the 32-bit env index of the random streams')
        if env_episode_offsets is not None:
            if not isinstance(self.spec.episode_time_steps, int):
                raise ValueError('env_episode_offsets needs an integer episode_time_steps (schema or kwarg)')
        self.layout = None
        if observations in ('tensor', 'compact'):
            from .observations import ObservationLayout
            self.layout = ObservationLayout(self.spec, observation_mode, normalize_observations, reference_quirks)
        rf_cls = resolve_reward(self.spec.reward_function.get('type'))
        kind = getattr(rf_cls, 'device_kind', None)
        self._comfort = kind == 'comfort'
        self._rf_attrs = dict(self.spec.reward_function.get('attributes') or {})
        # SolarPenaltyAndComfortReward (reward_function.py:336-386): coefficient-weighted sum of the SolarPenaltyReward the
        # energy step writes and the ComfortReward the LSTM stage writes -- combined on the device, no extra kernel of ours
        self._combo = None
        if rf_cls.__name__ == 'SolarPenaltyAndComfortReward' and rf_cls.__module__.endswith('reward_function'):
            if not any(b.is_dynamics for b in self.spec.buildings):
                raise NotImplementedError('SolarPenaltyAndComfortReward needs the LSTM temperature stage (dynamics buildings)')
            self._combo = tuple(float(c) for c in (self._rf_attrs.get('coefficients') or (1.0, 1.0)))
            self._comfort, kind = True, abi.CLR_SOLAR_PENALTY
        elif self._comfort:
            kind = abi.CLR_DEFAULT                      # the energy step still writes net / district sums
        # User plugins (the reference's RewardFunction plugin surface, reward_function.py:65-88, for a batch): a class without a
        # fused epilogue runs after the step kernel through `calculate_batch(planes)`, torch on the device -- see `_reward_planes`
        self._plugin = None
        if kind is None and self._combo is None:
            if not callable(getattr(rf_cls, 'calculate_batch', None)):
                raise NotImplementedError(
                    f'{rf_cls.__name__} has neither a fused device epilogue nor a calculate_batch(planes) method: give it one '
                    "(planes: dict of [n_bldg, n_envs] device tensors keyed by the reference's observation names -> reward [n_bldg, n_envs] "
                    "or [n_envs]), or use RewardFunction | MARL | IndependentSACReward | SolarPenaltyReward | ComfortReward")
            self._plugin = rf_cls(self.get_metadata(), **self._rf_attrs)
            self._plugin.env_metadata = self.get_metadata()            # set after construction too, like the reference (citylearn.py:243)
            kind = abi.CLR_DEFAULT                                      # the step kernel still writes net / district sums
        self.reward_name = {v: k for k, v in REWARD_KINDS.items()}[kind]
        self.reward_exponent = float((self.spec.reward_function.get('attributes') or {}).get('exponent') or 1.0)
        self._episode = -1
        low, high = self.spec.action_limits()
        self.action_low = torch.from_numpy(low).to(self.device)
        self.action_high = torch.from_numpy(high).to(self.device)
        self.reset()

    def get_metadata(self) -> Mapping[str, Any]:
        """`env_metadata` of the reward function (citylearn.py:243, 897-937): static facts of the district."""
        sp = self.spec
        return {
            'central_agent': self.central_agent, 'shared_observations': list(getattr(sp, 'shared_observations', []) or []),
            'seconds_per_time_step': sp.seconds_per_time_step, 'random_seed': sp.random_seed, 'n_envs': self.n_envs,
            'buildings': [{
                'name': b.name,
                'cooling_storage': {'capacity': b.cooling_storage.capacity}, 'heating_storage': {'capacity': b.heating_storage.capacity},
                'dhw_storage': {'capacity': b.dhw_storage.capacity},
                'electrical_storage': {'capacity': b.electrical_storage.capacity, 'nominal_power': b.electrical_storage.nominal_power},
                'cooling_device': {'nominal_power': b.cooling_device.nominal_power},
                'heating_device': {'nominal_power': b.heating_device.nominal_power},
                'dhw_device': {'nominal_power': b.dhw_device.nominal_power}, 'pv': {'nominal_power': b.pv_nominal_power},
                'action_metadata': dict(b.action_metadata), 'observation_metadata': dict(b.observation_metadata),
            } for b in sp.buildings],
        }

    def _reward_planes(self, t: int) -> Dict[str, torch.Tensor]:
        """What a batched reward plugin sees of step `t` (just computed): the reference's reward-observation keys
        (`Building.observations(include_all=True)`, building.py:1336-1481) as device tensors -- env-dependent ones ``[n_bldg, n_envs]``,
        exogenous ones ``[n_bldg, 1]`` (or ``[n_bldg, n_envs]`` with per-env-block episode windows), broadcastable against each other."""
        e = self.engine
        if e.env_row0 is None:
            row = self._exo[t][:, None, :]                                            # [B, 1, NF]
        else:
            block = torch.arange(e.n_env, device=self.device) // abi.CL_ROW0_BLOCK
            row = self._exo[e.env_row0.long() + t][block].permute(1, 0, 2)            # [B, E, NF]
        col = lambda c: row[:, :, c]
        net = e.out_bldg[abi.CLO_NET]
        planes = {
            'net_electricity_consumption': net,
            'net_electricity_consumption_cost': net * col(abi.CLT_PRICE),
            'net_electricity_consumption_emission': torch.clamp(net * col(abi.CLT_CARBON), min=0.0),
            'electrical_storage_soc': e.state[abi.CLS_B_SOC], 'cooling_storage_soc': e.state[abi.CLS_CS_SOC],
            'heating_storage_soc': e.state[abi.CLS_HS_SOC], 'dhw_storage_soc': e.state[abi.CLS_DS_SOC],
            'electricity_pricing': col(abi.CLT_PRICE), 'carbon_intensity': col(abi.CLT_CARBON),
            'non_shiftable_load': col(abi.CLT_NSL), 'solar_generation': -col(abi.CLT_SOLAR),
            'outdoor_dry_bulb_temperature': col(abi.CLT_T_OUT), 'hvac_mode': col(abi.CLT_HVAC_MODE), 'power_outage': col(abi.CLT_OUTAGE),
        }
        if e.detail is True:                         # (every detail plane, not the 'min' subset of the KPI pass / the LSTM stage)
            planes.update({'cooling_demand': e.out_bldg[abi.CLO_COOL_DEM], 'heating_demand': e.out_bldg[abi.CLO_HEAT_DEM],
                           'dhw_demand': e.out_bldg[abi.CLO_DHW_DEM], 'cooling_electricity_consumption': e.out_bldg[abi.CLO_C_COOL],
                           'heating_electricity_consumption': e.out_bldg[abi.CLO_C_HEAT], 'dhw_electricity_consumption': e.out_bldg[abi.CLO_C_DHW]})
        else:
            planes.update({'cooling_demand': col(abi.CLT_COOL_DEM), 'heating_demand': col(abi.CLT_HEAT_DEM), 'dhw_demand': col(abi.CLT_DHW_DEM)})
        if self.stage is not None:
            from .dynamics import PRE_BAND, PRE_CSP, PRE_HSP, PRE_OCC
            pre = self.stage.dyn_pre
            prow = pre[t][:, None, :] if e.env_row0 is None else pre[e.env_row0.long() + t][block].permute(1, 0, 2)
            planes.update({'indoor_dry_bulb_temperature': self.stage.indoor_temp,
                           'indoor_dry_bulb_temperature_cooling_set_point': prow[:, :, PRE_CSP],
                           'indoor_dry_bulb_temperature_heating_set_point': prow[:, :, PRE_HSP],
                           'occupant_count': prow[:, :, PRE_OCC], 'comfort_band': prow[:, :, PRE_BAND]})
        return planes

    @property
    def district_spec(self) -> DistrictSpec:
        """The loaded district (the name `CityLearnEnv` uses: there `spec` belongs to gymnasium when it is installed)."""
        return self.spec

    @property
    def n_act_cols(self) -> int:
        return self.engine.n_act_cols

    @property
    def n_bldg(self) -> int:
        return self.engine.n_bldg

    @property
    def time_step(self) -> int:
        return self._t

    @property
    def time_steps(self) -> int:
        return self.engine.n_steps

    @property
    def terminated(self) -> bool:
        return self._t == self.time_steps - 1

    def reset(self, seed: Optional[int] = None) -> Tuple[Dict[str, torch.Tensor], dict]:
        self._episode += 1
        self._reset_seed = seed
        if self._plugin is not None and callable(getattr(self._plugin, 'reset', None)):
            self._plugin.reset()                                  # RewardFunction.reset (citylearn.py:1846)
        n_steps, row0 = None, None
        if self.env_episode_offsets is None:
            window = self.spec.episode_window(self._episode, seed)
            redrawn = any(b.outage.simulate and b.outage.stochastic and b.outage.random_seed is None for b in self.spec.buildings)
            if getattr(self, 'engine', None) is not None and (self.tables.start, self.tables.end) == tuple(window) and not redrawn:
                # same episode window as the last one (the common case: one split, or `episode_time_steps` unset): the packed
                # tables, device copies and every state / output plane are reused -- only the state is re-initialised
                if self.engine.flex is not None:
                    self.engine.flex.seed = ((self.spec.random_seed if self._ev_seed is None else self._ev_seed) + self._episode) & (2 ** 64 - 1)
                self.engine.reset()
                if self.stage is not None:
                    self.stage.reset()
                self._t = 0
                return self._obs(), {}
            self.tables = self.spec.episode_tables(self._episode, seed, reward_exponent=self.reward_exponent)
        else:
            sp = self.spec
            self.tables = sp.episode_tables(reward_exponent=self.reward_exponent,
                                            window=(sp.simulation_start_time_step, sp.simulation_end_time_step))
            n_steps = int(sp.episode_time_steps)
            row0 = self._block_offsets(n_steps, self.tables.n_steps, seed)
        self.episode_row0 = row0
        obs_tables = self.layout.episode(self.tables, reset_table=row0 is not None) if self.layout is not None else None
        self.engine = StepEngine(self.tables, self.n_envs, device=str(self.device), reward=self.reward_name,
                                 t0_quirk=self.reference_quirks, kpi=self.kpi, n_steps=n_steps, env_row0=row0,
                                 # (a batched reward plugin sees the reference's full reward-observation key set: detail planes on)
                                 # ... and the LSTM stage alone only reads the delivered-demand planes: detail 'min'
                                 detail=True if (bool(obs_tables and obs_tables.needs_detail) or self._plugin is not None)
                                 else ('min' if any(b.is_dynamics for b in self.spec.buildings) else False),
                                 ev_reward_weights=self._rf_attrs.get('weights'), ev_drift=self._ev_drift, central_agent=self.central_agent,
                                 ev_penalty_coefficient=self._rf_attrs.get('charging_constraint_penalty_coefficient') or 1.0,
                                 ev_seed=(self.spec.random_seed if self._ev_seed is None else self._ev_seed) + self._episode, env_offset=self.env_offset,
                                 f64_maps=self.f64_maps)
        self.stage = None
        if any(b.is_dynamics for b in self.spec.buildings):
            from .dynamics import LSTMStage
            a = self._rf_attrs if self._comfort else {}
            self.stage = LSTMStage(self.spec, self.tables, self.engine, a.get('band'), a.get('lower_exponent') or 2.0,
                                   a.get('higher_exponent') or 2.0, kpi=self.kpi)
        self._t = 0
        self._exo = self.engine.ts            # [T, B, CL_NF] on device: exogenous values per (t, building)
        self.writer = None
        if obs_tables is not None:
            from .observe import ObservationWriter
            if self._compact:
                dep_tables, cols = obs_tables.compact()
                self._shared_rows = torch.from_numpy(np.ascontiguousarray(obs_tables.table, dtype=np.float32)).to(self.device)
                self._shared_reset = None if obs_tables.reset_table is None else \
                    torch.from_numpy(np.ascontiguousarray(obs_tables.reset_table, dtype=np.float32)).to(self.device)
                self._dep_cols = torch.from_numpy(cols.astype(np.int64)).to(self.device)
                self.writer = ObservationWriter(self.engine, dep_tables, self.stage) if len(cols) else None
            else:
                self.writer = ObservationWriter(self.engine, obs_tables, self.stage)
        return self._obs(), {}

    # ---- checkpoint / restore -------------------------------------------------------------------------------------------------------
    def state_dict(self) -> dict:
        """A complete checkpoint of the running episode as plain tensors and scalars (`torch.save`-able): the engine's planes, KPI accumulators
        and flexible-load state (`StepEngine.state_dict`), the LSTM stage's rings and hidden state, and the episode bookkeeping -- episode
        number, the seed `reset` was called with, the time step, the episode window and the per-block offsets.  The reference's counterpart is
        pickling the whole env (citylearn/__main__.py:291-299); here the tables are rebuilt from the schema and only what moves is saved."""
        return {'format': 1, 'episode': self._episode, 'reset_seed': self._reset_seed, 't': int(self._t),
                'window': (int(self.tables.start), int(self.tables.end)),
                'episode_row0': None if self.episode_row0 is None else np.asarray(self.episode_row0).copy(),
                'engine': self.engine.state_dict(), 'stage': None if self.stage is None else self.stage.state_dict()}

    def load_state_dict(self, sd: Mapping[str, Any]) -> None:
        """Restore :meth:`state_dict` into this env (constructed on the same schema with the same arguments).  If the checkpoint belongs to
        another episode -- another window of the data, other outage draws, other block offsets -- that episode is rebuilt first (`reset` with the
        saved episode number and seed); then every carried tensor is copied in place.  The next `step` continues bit-identically."""
        if sd.get('format') != 1:
            raise ValueError(f"checkpoint format {sd.get('format')!r}, this build reads 1")
        same_window = (int(self.tables.start), int(self.tables.end)) == tuple(sd['window'])
        same_rows = (self.episode_row0 is None) == (sd['episode_row0'] is None) and \
                    (self.episode_row0 is None or np.array_equal(np.asarray(self.episode_row0), sd['episode_row0']))
        if self._episode != sd['episode'] or not same_window or not same_rows:
            self._episode = int(sd['episode']) - 1
            self.reset(sd['reset_seed'])
            if (int(self.tables.start), int(self.tables.end)) != tuple(sd['window']):
                raise ValueError(f"episode {sd['episode']} of this env covers rows {(self.tables.start, self.tables.end)}, the checkpoint's covered {tuple(sd['window'])}: "
                                 'not the same schema / episode split')
        self._reset_seed = sd['reset_seed']
        self.engine.load_state_dict(sd['engine'])
        if (self.stage is None) != (sd['stage'] is None):
            raise ValueError('checkpoint and env disagree about the LSTM temperature stage')
        if self.stage is not None:
            self.stage.load_state_dict(sd['stage'])
        self._t = int(sd['t'])
        self.engine.t = self._t

    def _block_offsets(self, n_steps: int, n_rows: int, seed: Optional[int]) -> np.ndarray:
        n_blocks = -(-self.n_envs // abi.CL_ROW0_BLOCK)
        latest = n_rows - n_steps
        if latest < 0:
            raise ValueError(f'episode_time_steps={n_steps} exceeds the {n_rows} rows of the simulation period')
        mode = self.env_episode_offsets
        if isinstance(mode, str):
            if mode == 'rolling':                         # block g starts g episodes (or g rows when they run out) further
                stride = n_steps if n_blocks * n_steps <= latest + n_steps else max(1, latest // max(n_blocks - 1, 1))
                return (np.arange(n_blocks) * stride) % (latest + 1)
            if mode == 'random':
                s = self.spec.random_seed if seed is None else seed
                # the episode enters the seed additively (a product would make every episode identical for seed 0, the
                # loader's default when neither the schema nor the caller gives one)
                return np.random.RandomState(np.random.SeedSequence([int(s) & 0xFFFFFFFF, self._episode]).generate_state(1)[0]).randint(0, latest + 1, size=n_blocks)
            raise ValueError("env_episode_offsets must be None, 'rolling', 'random' or an array")
        row0 = np.asarray(mode, dtype=np.int64).reshape(-1)
        if row0.shape[0] != n_blocks:
            raise ValueError(f'env_episode_offsets needs {n_blocks} entries (one per {abi.CL_ROW0_BLOCK} envs)')
        return row0

    @property
    def observation_names(self):
        if self.layout is None:
            raise RuntimeError("construct VectorCityLearnEnv(..., observations='tensor') for named observation columns")
        return self.layout.names

    @property
    def observation_space(self):
        from .spaces import Box
        return [Box(low=lo, high=hi, dtype=np.float32) for lo, hi in self.layout.space()]

    def _obs(self, dep=None):
        e = self.engine
        if self._compact and self.layout is not None:
            row = min(self._t, e.n_steps - 1)
            table = self._shared_reset if (row == 0 and self._shared_reset is not None) else self._shared_rows
            shared = table[row] if e.env_row0 is None else table[e.env_row0.long() + row]          # [n_obs] or [n_blocks, n_obs]
            if dep is None:                                  # (step() hands over what cl_step_observe_f32 already wrote)
                dep = self.writer.write(row) if self.writer is not None else torch.zeros((e.n_env, 0), device=self.device)
            return {'shared': shared, 'dependent': dep, 'columns': self._dep_cols}
        if self.writer is not None:
            return self.writer.write(min(self._t, e.n_steps - 1))
        t_row = min(self._t, e.n_steps - 1)
        # per-env-block episode windows: one exogenous row per block, [n_blocks, n_bldg, CL_NF]
        exo = self._exo[t_row] if e.env_row0 is None else self._exo[e.env_row0.long() + t_row]
        return {'exogenous': exo,
                'electrical_storage_soc': e.state[abi.CLS_B_SOC], 'cooling_storage_soc': e.state[abi.CLS_CS_SOC],
                'heating_storage_soc': e.state[abi.CLS_HS_SOC], 'dhw_storage_soc': e.state[abi.CLS_DS_SOC],
                'net_electricity_consumption': e.out_bldg[abi.CLO_NET],
                **({'electric_vehicle_soc': e.ev_state[0]} if e.flex is not None else {}),
                **({'indoor_dry_bulb_temperature': self.stage.indoor_temp} if getattr(self, 'stage', None) is not None else {})}

    def step(self, actions: torch.Tensor):
        if self.terminated:
            raise RuntimeError('episode has terminated: call reset()')
        e = self.engine
        if actions.shape == (e.n_env, e.n_act_cols) and e.n_env != e.n_act_cols:
            actions = actions.t()                       # strided view, no copy
        dep = None
        if self._compact and self.writer is not None and self.stage is None:
            dep = e.step_observe(actions, self.writer, self._t)      # the dependent columns of the next observation, same launch where possible
        else:
            e.step(actions, self._t)
        if self.stage is not None:
            self.stage.step(self._t)                    # indoor temperature (+ ComfortReward) of this step
        self._t += 1
        if self._plugin is not None:
            r = self._plugin.calculate_batch(self._reward_planes(self._t - 1))
            if r.dim() == 2 and r.shape[0] == e.n_bldg:
                reward = (r.expand(e.n_bldg, e.n_env).sum(dim=0) if self.central_agent else r.expand(e.n_bldg, e.n_env))
            elif r.dim() == 1 and r.shape[0] == e.n_env:
                reward = r
            else:
                raise ValueError(f'calculate_batch returned shape {tuple(r.shape)}; expected [n_bldg, n_envs] (or broadcastable) or [n_envs]')
        elif self._combo is not None:
            # float32 like the reference's np.array(..., dtype='float32') (reward_function.py:383-386)
            r = self._combo[0] * e.reward_bldg + self._combo[1] * self.stage.comfort
            reward = r.sum(dim=0) if self.central_agent else r
        elif self._comfort and self.stage is not None:
            reward = self.stage.comfort.sum(dim=0) if self.central_agent else self.stage.comfort
        else:
            reward = e.district_reward if self.central_agent else e.reward_bldg
        return self._obs(dep), reward, self.terminated, False, {}

    def capture(self, actions: torch.Tensor) -> 'CapturedSteps':
        """`step` as hipGraph replays.  `actions` is a persistent float32 buffer (``[n_act_cols, n_envs]`` or the transposed policy
        layout) the caller overwrites before every `CapturedSteps.step()`; the launches of a step (energy step, flexible loads, LSTM stage,
        observation epilogue, reward) are captured once per time step of the episode -- the step index and the table row are kernel
        arguments -- and replayed afterwards, so a Python RL loop pays one graph launch per step instead of the ctypes calls of an
        eager step (scripts/env_step_bench.py: eager vs captured us per step).  The returned observation / reward tensors are the
        engine's own buffers (or graph-owned ones): valid until the next step."""
        return CapturedSteps(self, actions)

    def capture_rollout(self, policy, k_steps: int, keep_rewards: bool = True) -> 'CapturedRollout':
        """``k_steps`` x (policy, `step`) as ONE hipGraph replay -- the closed-loop counterpart of :meth:`rollout` for a policy written in
        torch.  ``policy(obs, i)`` is called while the graph is being captured, once per step ``i`` of the chunk, with the observation
        `step` would hand out, and returns that step's actions (float32 ``[n_act_cols, n_envs]`` or transposed, on the env's device);
        whatever device work it enqueues on the current stream (an MLP, sampling with torch's graph-safe generator, writes into its own
        trajectory buffers) becomes part of the graph, between the env's kernels.  A Python RL loop then pays one graph launch per
        ``k_steps`` env steps instead of one trip through Python, ctypes and the HIP launch path per step: what `bench.py` times, behind
        the user-level API (scripts/env_step_bench.py).  See `CapturedRollout`."""
        return CapturedRollout(self, policy, int(k_steps), keep_rewards)

    def rollout(self, k_steps: int, actions: Optional[torch.Tensor] = None, seed: int = 0) -> torch.Tensor:
        """Advance ``k_steps`` steps without returning to Python in between (`StepEngine.rollout`: one fused launch, or a launch
        sequence for districts with flexible loads) with open-loop ``actions`` ``[k_steps, n_act_cols, n_envs]`` or the uniform
        random policy keyed by ``seed`` (the device analogue of `Agent.predict`, agents/base.py:188-209).  Returns the district
        reward summed over those steps, ``[n_envs]``.  Streaming KPIs (``kpi=True``) are updated after every step.  Not available with
        the LSTM temperature stage or a batched reward plugin, which run their own code between steps -- use :meth:`step` there."""
        if self.stage is not None or self._plugin is not None:
            raise NotImplementedError('rollout() needs a district without the LSTM temperature stage and a fused reward; use step()')
        if self._t + k_steps > self.time_steps - 1:
            raise RuntimeError(f'{k_steps} steps from t={self._t} run past the episode end ({self.time_steps - 1} steps)')
        e = self.engine
        if actions is None and e.act_low is None:
            e.set_action_limits(self.action_low.cpu().numpy(), self.action_high.cpu().numpy())
        ret = torch.zeros(self.n_envs, dtype=torch.float32, device=self.device)
        e.rollout(k_steps, actions=actions, seed=seed, ret_env=ret, t0=self._t)
        self._t += k_steps
        return ret

    def evaluate(self):
        """Per-env KPI ratios of `CityLearnEnv.evaluate` (citylearn.py:1136-1323) from the on-device streaming
        accumulators (construct with ``kpi=True``).  Returns ``(building, district)`` dicts of tensors."""
        if not self.kpi:
            raise RuntimeError('construct VectorCityLearnEnv(..., kpi=True) to accumulate KPIs on the device')
        from .kpi import finalize_comfort, finalize_streaming
        from .schema import EpisodeTables

        def block(tables, sl):
            """KPIs of the envs `sl`, which all replay the episode window `tables`."""
            nxt_e = nxt_o = None
            if self._t < self.time_steps:
                row = tables.start + self._t
                nxt_e = np.array([float(b.series['cooling_demand'][row]) + float(b.series['heating_demand'][row])
                                  + float(b.series['dhw_demand'][row]) + float(b.series['non_shiftable_load'][row])
                                  for b in self.spec.buildings])
                nxt_o = (tables.outage[self._t] != 0).astype(np.float64)
            building, district = finalize_streaming(self.engine.kpi_bldg[:, :, sl], self.engine.kpi_env[:, sl], self._t, self.time_steps,
                                                    nxt_e, nxt_o, shared_baseline=self.engine.kpi_shared_baseline)
            if self.stage is not None and self.stage.kpi_comfort is not None:
                comfort = finalize_comfort(self.stage.kpi_comfort[:, :, sl], self.spec, tables, self._t, self.stage.kpi_band)
                building.update(comfort)
                for name, v in comfort.items():
                    district[name] = torch.nanmean(v, dim=0)
            return building, district

        if self.episode_row0 is None:
            return block(self.tables, slice(None))
        # per-env-block episode windows: finalise every block against its own window of the tables
        parts = []
        K, tab = self.time_steps, self.tables
        for g, o in enumerate(self.episode_row0):
            o = int(o)
            cut = EpisodeTables(params=tab.params, ts=tab.ts[o:o + K], start=tab.start + o, end=tab.start + o + K - 1,
                                outage=tab.outage[o:o + K])
            parts.append(block(cut, slice(g * abi.CL_ROW0_BLOCK, min((g + 1) * abi.CL_ROW0_BLOCK, self.n_envs))))
        building = {k: torch.cat([p[0][k] for p in parts], dim=-1) for k in parts[0][0]}
        district = {k: torch.cat([p[1][k] for p in parts], dim=-1) for k in parts[0][1]}
        return building, district

    def materialize(self, obs: Mapping[str, torch.Tensor]) -> torch.Tensor:
        """The ``[n_envs, n_obs]`` observation matrix of a ``'compact'`` observation (what ``observations='tensor'`` returns)."""
        shared = obs['shared']
        if shared.dim() == 2:                                       # one row per env block (per-env-block episode windows)
            block = torch.arange(self.n_envs, device=self.device) // abi.CL_ROW0_BLOCK
            full = shared[block].clone()
        else:
            full = shared.unsqueeze(0).repeat(self.n_envs, 1)
        full[:, obs['columns']] = obs['dependent']
        return full

    def sample_actions(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """Uniform random actions inside the action space (the device analogue of `Agent.predict`, agents/base.py:188-209)."""
        u = torch.rand((self.n_act_cols, self.n_envs), device=self.device, generator=generator)
        return self.action_low[:, None] + u * (self.action_high - self.action_low)[:, None]


class CapturedSteps:
    """`VectorCityLearnEnv.step` through hipGraph replay (see `VectorCityLearnEnv.capture`).  One graph per time step of the episode,
    captured the first time the step is reached (the capture itself does not advance the env; the replay right after it does) and reused
    by every later episode that replays the same window -- a `reset()` that rebuilds the engine (another episode window) drops them."""

    def __init__(self, env: VectorCityLearnEnv, actions: torch.Tensor):
        e = env.engine
        if actions.dtype != torch.float32 or actions.device != e.device or tuple(actions.shape) not in ((e.n_act_cols, e.n_env), (e.n_env, e.n_act_cols)):
            raise ValueError(f'actions must be a float32 [{e.n_act_cols}, {e.n_env}] (or transposed) tensor on {e.device}')
        self.env, self.actions = env, actions
        self._key = self._state_key()
        self._graphs = {}
        self._stream = torch.cuda.Stream(device=e.device)

    def _state_key(self):
        e = self.env.engine
        return (id(e), None if e.flex is None else int(e.flex.seed))

    def step(self):
        """Advance by one step with the current content of the action buffer; returns what `VectorCityLearnEnv.step` returns."""
        env = self.env
        if self._state_key() != self._key:
            # reset() built a new engine (the captured launches point at freed buffers) or moved a by-value kernel argument (the drift
            # seed of a district with EVs advances with the episode): capture again
            self._graphs.clear()
            self._key = self._state_key()
        if env.terminated:
            raise RuntimeError('episode has terminated: call reset()')
        t = env._t
        hit = self._graphs.get(t)
        if hit is None:
            self._stream.wait_stream(torch.cuda.current_stream(env.engine.device))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=self._stream):
                out = env.step(self.actions)                     # recorded, not executed: the state on the device is untouched
            env._t = t                                           # (env.step advanced the host-side clock)
            hit = self._graphs[t] = (graph, out)
        graph, out = hit
        graph.replay()
        env._t = t + 1
        env.engine.t = t + 1
        return out[0], out[1], env.terminated, False, {}


class CapturedRollout:
    """`VectorCityLearnEnv.capture_rollout`: chunks of ``k_steps`` closed-loop env steps, one hipGraph per chunk start ``t0`` (the step
    index and the table row are kernel arguments), captured the first time the env stands at ``t0`` -- the capture does not advance
    the env, the replay right after it does -- and reused by every later episode over the same window.

    `run()` returns ``(observation, rewards, terminated)``: the observation after the chunk's last step (the env's own buffers, valid
    until the next step), and ``rewards`` ``[k_steps, ...]`` = what `step` returned at each step (a graph-owned buffer, overwritten by
    the next `run()` from the same ``t0``; None with ``keep_rewards=False`` for policies that record what they need themselves)."""

    def __init__(self, env: VectorCityLearnEnv, policy, k_steps: int, keep_rewards: bool = True):
        if k_steps < 1:
            raise ValueError('k_steps must be at least 1')
        self.env, self.policy, self.k_steps, self.keep_rewards = env, policy, k_steps, keep_rewards
        self._key = self._state_key()
        self._graphs = {}
        self._stream = torch.cuda.Stream(device=env.engine.device)

    def _state_key(self):
        e = self.env.engine
        return (id(e), None if e.flex is None else int(e.flex.seed))

    def run(self, observation=None):
        """Advance the env by ``k_steps`` steps.  ``observation``: ignored after the first capture of a chunk (the graph reads the env's
        own observation buffers); at capture time it defaults to the env's current observation."""
        env, k = self.env, self.k_steps
        if self._state_key() != self._key:              # another engine / EV drift seed: the recorded launches are stale
            self._graphs.clear()
            self._key = self._state_key()
        if env.terminated:
            raise RuntimeError('episode has terminated: call reset()')
        t0 = env._t
        if t0 + k > env.time_steps - 1:
            raise RuntimeError(f'{k} steps from t={t0} run past the episode end ({env.time_steps - 1} steps): finish the episode with step()')
        hit = self._graphs.get(t0)
        if hit is None:
            dev = env.engine.device
            self._stream.wait_stream(torch.cuda.current_stream(dev))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=self._stream):
                obs = env._obs() if observation is None else observation
                rewards = None
                for i in range(k):
                    obs, reward, _, _, _ = env.step(self.policy(obs, i))
                    if self.keep_rewards:
                        if rewards is None:
                            rewards = torch.empty((k,) + tuple(reward.shape), dtype=reward.dtype, device=dev)
                        rewards[i].copy_(reward)
            env._t = t0                                  # (recorded, not executed)
            hit = self._graphs[t0] = (graph, obs, rewards)
        graph, obs, rewards = hit
        graph.replay()
        env._t = t0 + k
        env.engine.t = t0 + k
        return obs, rewards, env.terminated
