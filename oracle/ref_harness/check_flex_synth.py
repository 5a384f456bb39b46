"""Pins oracle/flex_oracle.py on synthetic charger schedules (tests/flex_synth.py) by running the REFERENCE on the same files
in this container: EV SoC series, charger consumption and building nets must agree bit for bit.  Not a test (the reference is
not available on the GPU box); run by hand: python oracle/ref_harness/check_flex_synth.py [seed ...]"""
import json
import random as py_random
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / 'tests')); sys.path.insert(0, str(Path(__file__).resolve().parent))
import ref_env  # noqa: E402
from flex_synth import make  # noqa: E402


def main():
    seeds = [int(x) for x in sys.argv[1:]] or [1, 2, 3]
    ref_env.setup_reference()
    from citylearn.citylearn import CityLearnEnv
    from citylearn_amd.schema import load_district
    from oracle.flex_oracle import FlexDistrictOracle
    src = REPO / 'tests' / 'golden' / 'g2022_evs' / 'dataset'
    for seed in seeds:
        schema = make(src, Path('/tmp') / f'flex_synth_{seed}', seed, curves=seed % 2 == 0)      # even seeds: charger efficiency curves
        py_random.seed(seed); np.random.seed(seed)
        env = CityLearnEnv(str(schema))
        env.reset()
        spec = load_district(str(schema))
        for ev, ref_ev in zip(spec.electric_vehicles, env.electric_vehicles):
            ev.battery.initial_soc = float(ref_ev.battery.initial_soc)
        tab = spec.episode_tables(0)
        np.random.seed(seed)
        # the env constructor + reset consumed no np.random draws before the first step (drift draws start in next_time_step)
        o = FlexDistrictOracle(spec, tab, 1, reward='Electric_Vehicles_Reward_Function')
        o.reset()
        low = np.concatenate([b.action_space.low for b in env.buildings]); high = np.concatenate([b.action_space.high for b in env.buildings])
        sizes = [b.action_space.shape[0] for b in env.buildings]
        rng = np.random.RandomState(seed + 100)
        worst = {}
        K = 200
        for t in range(K):
            a = rng.uniform(low, high).astype('float32')
            a[rng.uniform(size=len(a)) < 0.15] = 0.0
            acts, p = [], 0
            for s in sizes:
                acts.append([float(x) for x in a[p:p + s]]); p += s
            st = np.random.get_state()
            _, r, *_ = env.step(acts)
            np.random.set_state(st)                     # the oracle replays exactly the draws the reference made in this step
            out = o.step(a[:, None])
            got = {'ev_soc': out['ev_soc'][:, 0], 'net': out['net'][:, 0], 'reward': out['reward'][:, 0],
                   'charger': out['charger_consumption'][:, 0]}
            ref = {'ev_soc': np.array([ev.battery.soc[t] for ev in env.electric_vehicles], dtype='float32'),
                   'net': np.array([b._Building__net_electricity_consumption[t] for b in env.buildings], dtype='float32'),
                   'reward': np.array(r, dtype='float64'),
                   'charger': np.array([c.electricity_consumption[t] for b in env.buildings for c in b.electric_vehicle_chargers], dtype='float32')}
            for k in got:
                d = float(np.max(np.abs(got[k].astype(np.float64) - ref[k].astype(np.float64))))
                worst[k] = max(worst.get(k, 0.0), d)
        print(f'seed {seed}: worst |oracle - reference| over {K} steps:', json.dumps(worst))


if __name__ == '__main__':
    main()
