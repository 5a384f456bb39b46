"""Synthetic large districts for scaling runs (BASELINE config 4): a loaded district tiled to N buildings with
device sizes jittered by a seeded +-`jitter` factor.  The per-building data series are shared (not copied)."""
from __future__ import annotations

import copy
from dataclasses import replace

import numpy as np

from .schema import DistrictSpec


def tile_district(spec: DistrictSpec, n_buildings: int, seed: int = 4, jitter: float = 0.10) -> DistrictSpec:
    rng = np.random.RandomState(seed)
    base = spec.buildings
    out = []
    for i in range(n_buildings):
        src = base[i % len(base)]
        f = lambda: float(1.0 + jitter * (2.0 * rng.rand() - 1.0))
        b = copy.copy(src)                         # shallow: `series` dict / arrays are shared
        b.name = f'{src.name}_x{i // len(base):03d}' if i >= len(base) else src.name
        es = copy.copy(src.electrical_storage)
        es.capacity, es.nominal_power = es.capacity * f(), es.nominal_power * f()
        b.electrical_storage = es
        b.pv_nominal_power = src.pv_nominal_power * f()
        for key in ('cooling_storage', 'heating_storage', 'dhw_storage'):
            tank = copy.copy(getattr(src, key))
            tank.capacity = float(tank.capacity) * f()
            setattr(b, key, tank)
        for key in ('cooling_device', 'heating_device', 'dhw_device'):
            dev = copy.copy(getattr(src, key))
            dev.nominal_power = float(dev.nominal_power) * (1.0 + jitter * rng.rand())     # never undersized
            setattr(b, key, dev)
        b.action_metadata = dict(src.action_metadata)
        b.observation_metadata = dict(src.observation_metadata)
        out.append(b)
    return replace(spec, buildings=out)
