"""Small fixture `tests/golden/overrides.json`: what the REFERENCE does with constructor overrides on the committed mini
datasets -- episode windows across resets (EpisodeTracker, base.py:100-129), building subsets, central-agent switches,
(in)active observations / actions -- names, action bounds and the first returned observation.

TEST INFRASTRUCTURE ONLY.  One reference process per case (see gen_golden.py); run: python oracle/ref_harness/gen_overrides.py
"""
import json
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))

CASES = {
    'episodes_fixed': ('g2022_all', {'episode_time_steps': 100}),
    'episodes_rolling': ('g2022_all', {'episode_time_steps': 100, 'rolling_episode_split': True}),
    # the reference forwards the constructor's `random_episode_split` under the wrong name (citylearn.py:191 vs 2045): the
    # kwarg is silently ignored and only the schema's value counts
    'episodes_random_kwarg_ignored': ('g2022_all', {'episode_time_steps': 100, 'random_episode_split': True}),
    'episodes_random': ('g2022_all', {'episode_time_steps': 100}, {'random_episode_split': True, 'random_seed': 3}),
    'episodes_rolling_random': ('g2022_all', {'episode_time_steps': 100, 'rolling_episode_split': True}, {'random_episode_split': True, 'random_seed': 5}),
    'episodes_list': ('g2022_all', {'episode_time_steps': [[0, 47], [100, 299], [600, 719]]}),
    'window': ('g2022_all', {'simulation_start_time_step': 24, 'simulation_end_time_step': 407}),
    'subset_central': ('g2022_all', {'buildings': ['Building_2', 'Building_9', 'Building_17'], 'central_agent': True}),
    'decentral_2023': ('g2023_p2', {'central_agent': False}),
    'inactive': ('g2020_cz1', {'inactive_observations': ['hour', 'dhw_storage_soc'], 'inactive_actions': ['dhw_storage']}),
    'active_only': ('g2020_cz1', {'active_observations': ['hour', 'net_electricity_consumption', 'electrical_storage_soc'],
                                  'active_actions': ['electrical_storage']}),
    'no_outage': ('g2023_p2', {'simulate_power_outage': False}),
    'no_pv': ('g2022_all', {'solar_generation': False, 'buildings': ['Building_1', 'Building_3']}),
    # stochastic outage signals are redrawn every episode (building.py:2566-2594)
    'episodes_outage': ('g2023_p2', {'episode_time_steps': 240}),
    'episodes_outage_seeded': ('g2023_p2', {'episode_time_steps': 240, 'random_seed': 11}),
    # reset(seed=...) re-seeds the episode choice and the outage draws (citylearn.py:1855-1866)
    'reset_seeds': ('g2023_p2', {'episode_time_steps': 240}, {'random_episode_split': True}, [None, 5, 5, 9, None, 2, 13]),
}


def run_case(name: str):
    import numpy as np
    import ref_env
    ref_env.setup_reference()
    from citylearn.citylearn import CityLearnEnv
    fixture, kwargs, *rest = CASES[name]
    schema_overrides = rest[0] if rest else {}
    seeds = rest[1] if len(rest) > 1 else [None] * 7
    dataset_dir = REPO / 'tests' / 'golden' / fixture / 'dataset'
    if not dataset_dir.exists():                      # the 2022_phase_all sample lives in the package (citylearn_amd/data)
        dataset_dir = REPO / 'citylearn_amd' / 'data' / 'citylearn_challenge_2022_phase_all_720h'
    schema = str(dataset_dir / 'schema.json')
    if schema_overrides:                              # schema handed over as a dictionary with the overrides applied
        schema = {**json.loads(Path(schema).read_text()), **schema_overrides,
                  'root_directory': str(dataset_dir)}
    env = CityLearnEnv(schema, **kwargs)
    out = {'fixture': fixture, 'kwargs': kwargs, 'schema_overrides': schema_overrides, 'reset_seeds': seeds, 'central_agent': bool(env.central_agent),
           'building_names': [b.name for b in env.buildings], 'observation_names': env.observation_names,
           'action_names': env.action_names, 'time_steps': int(env.time_steps),
           'action_low': [s.low.tolist() for s in env.action_space], 'action_high': [s.high.tolist() for s in env.action_space],
           'obs_low': [s.low.astype(float).tolist() for s in env.observation_space],
           'obs_high': [s.high.astype(float).tolist() for s in env.observation_space], 'episodes': []}
    for ep in range(7):
        obs, _ = env.reset(seed=seeds[ep])
        tr = env.episode_tracker
        out['episodes'].append({'start': int(tr.episode_start_time_step), 'end': int(tr.episode_end_time_step),
                                'obs0': [[float(x) for x in o] for o in obs],
                                'outage': [np.flatnonzero(np.array(b._Building__power_outage_signal)).tolist() for b in env.buildings]})
        if ep == 0:
            rng = np.random.RandomState(7)
            acts = [[float(x) for x in rng.uniform(s.low, s.high).astype('float32')] for s in env.action_space]
            o1, r1, *_ = env.step(acts)
            out['first_step'] = {'actions': acts, 'reward': [float(x) for x in r1],
                                 'net': [float(b.net_electricity_consumption[0]) for b in env.buildings]}
    print(json.dumps(out))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--one':
        run_case(sys.argv[2])
        sys.exit(0)
    result = {}
    for name in CASES:
        p = subprocess.run([sys.executable, __file__, '--one', name], check=True, capture_output=True, text=True)
        result[name] = json.loads(p.stdout.strip().splitlines()[-1])
        print(name, 'ok', result[name]['episodes'][0]['start'], result[name]['episodes'][0]['end'])
    (REPO / 'tests' / 'golden' / 'overrides.json').write_text(json.dumps(result))
