"""Observation wrapper mirror of the reference's `citylearn/wrappers.py` for the hot path's caller side.

Only `NormalizedObservationWrapper` (wrappers.py:39-167) is provided: it is part of what an agent sees of the step
(SURVEY 8f-3).  The batched form is ``VectorCityLearnEnv(..., observations='tensor', normalize_observations=True)``
where the normalisation is folded into the tables `cl_observe_f32` reads; this class is the list-based single-district
surface.  The remaining reference wrappers (discretisation, stable-baselines3 adapters, ...) are agent-side utilities
outside the scope table.
"""
from __future__ import annotations

from typing import List

import numpy as np

from .observations import ObservationLayout
from .spaces import Box


class NormalizedObservationWrapper:
    """Periodic (sin / cos of hour, day_type, month) + min-max normalisation of the observations; same names, order,
    limits and values as the reference wrapper."""

    def __init__(self, env):
        self.env = env
        base = env.unwrapped
        self._layout = ObservationLayout(base.district_spec, base.observation_mode, True, base.reference_quirks)
        self._episode = None

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def shared_observations(self) -> List[str]:
        return list(self._layout.shared)

    @property
    def observation_names(self) -> List[List[str]]:
        return self._layout.names

    @property
    def observation_space(self) -> List[Box]:
        return [Box(low=lo, high=hi, dtype=np.float32) for lo, hi in self._layout.space()]

    def _tables(self):
        base = self.env.unwrapped
        if self._episode != (base.episode, id(base._tables)):
            self._obs_tables = self._layout.episode(base._tables)
            self._episode = (base.episode, id(base._tables))
        return self._obs_tables

    def observation(self, observations=None) -> List[List[float]]:
        """Normalised observations of the env's current time step (the argument is ignored, like the reference, which
        re-reads the buildings)."""
        base = self.env.unwrapped
        tabs = self._tables()
        t = base.time_step
        # same rule as CityLearnEnv._observation_vector: the table row is complete only when no column depends on the env (the
        # charging headroom / violation columns do in every mode: they are attributes the last apply_actions overwrote, not series)
        if t == 0 or (base.observation_mode == 'reference' and tabs.n_dependent == 0):
            v = tabs.table[t]
        else:
            extra = None if base._engine.flex is None else base._engine.flex_out[:, :, 0].cpu().numpy()
            v = tabs.host_row(t, base._last_state, base._last_out, base._last_temps, extra)
        return [v[s].tolist() for s in self._layout.agent_slices]

    def reset(self, **kwargs):
        obs, info = self.env.reset(**kwargs)
        return self.observation(obs), info

    def step(self, actions):
        obs, reward, terminated, truncated, info = self.env.step(actions)
        return self.observation(obs), reward, terminated, truncated, info
