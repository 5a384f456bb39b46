"""Helpers to read the committed golden fixtures (tests/golden/<name>/{dataset/,reference.npz})."""
import json
from functools import lru_cache
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / 'golden'
_DATA = GOLDEN.parent.parent / 'citylearn_amd' / 'data'
PACKAGE_DATA = {'g2022_all': _DATA / 'citylearn_challenge_2022_phase_all_720h',                  # bench.py's headline / C2 / C5 tables
                'g2023_p2': _DATA / 'citylearn_challenge_2023_phase_2_local_evaluation_720h',    # bench.py --config C3
                'g2020_cz1': _DATA / 'citylearn_challenge_2020_climate_zone_1_744h'}             # bench.py --config C4 (device set)
FIXTURES = ('g2022_all', 'g2020_cz1', 'g2023_p2', 'g2022_p1_year', 'g2020_15min')
# dataset sweep: 95-step runs of the other dataset families (oracle/ref_harness/gen_golden.py)
SWEEP = ('s_baeda', 's_2021', 's_2020_cz3', 's_2023_p1', 's_2023_p3', 's_autosize',
         # round 3: every other dataset of the reference checkout that the reference itself can run here
         's_2020_cz2', 's_2022_p2', 's_2023_oe1', 's_2023_oe2', 's_2023_oe3', 's_2023_p32', 's_2023_p33')


class Golden:
    def __init__(self, name: str):
        self.name = name
        self.dir = GOLDEN / name
        # three samples ship with the package (bench.py / smoke() load them too); every other fixture keeps its own
        self.dataset_dir = self.dir / 'dataset' if (self.dir / 'dataset').exists() else PACKAGE_DATA[name]
        self.schema_path = str(self.dataset_dir / 'schema.json')
        self.ref = np.load(self.dir / 'reference.npz', allow_pickle=False)
        self.facts = json.loads(str(self.ref['facts']))
        self._obs = None

    @property
    def obs(self):
        """observations.npz: what the reference's reset()/step() return + NormalizedObservationWrapper view."""
        if self._obs is None:
            self._obs = np.load(self.dir / 'observations.npz', allow_pickle=False)
            self.obs_facts = json.loads(str(self._obs['facts']))
        return self._obs

    @property
    def reward_kind(self) -> str:
        k = self.facts['reward_type']
        return k if k in ('RewardFunction', 'MARL', 'IndependentSACReward', 'SolarPenaltyReward') else 'RewardFunction'

    def spec(self, schema_overrides=None, **kwargs):
        from citylearn_amd.schema import load_district
        if schema_overrides:            # schema as a dictionary with top-level keys replaced
            schema = {**json.loads(open(self.schema_path).read()), **schema_overrides, 'root_directory': str(self.dataset_dir)}
            return load_district(schema, **kwargs)
        if 'noise_seed' in self.facts:         # `noise_std` fixtures: the harness seeded numpy's global generator with this
            kwargs.setdefault('noise_seed', self.facts['noise_seed'])
        spec = load_district(self.schema_path, **kwargs)
        # EVs without an `initial_soc` get one draw of Python's global `random` in the reference (citylearn.py:2564), which
        # other libraries also consume during construction: the fixture records the values the reference ended up with
        for ev, fact in zip(spec.electric_vehicles, self.facts.get('electric_vehicles', [])):
            assert ev.name == fact['name']
            ev.battery.initial_soc = fact['initial_soc']
        return spec


@lru_cache(maxsize=None)
def golden(name: str) -> Golden:
    return Golden(name)
