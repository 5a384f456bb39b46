"""Synthetic charger schedules on top of the EV fixture's mini dataset: connection patterns the shipped dataset does not
contain (EVs swapping chargers, arrivals announced with and without an SoC, back-to-back connections, a charger that is
never used).  Used by the oracle-vs-device parity test; `oracle/ref_harness/check_flex_synth.py` runs the reference itself on
the same files to pin the oracle on them."""
import shutil
from pathlib import Path

import numpy as np

HEADER = ('electric_vehicle_charger_state,electric_vehicle_id,electric_vehicle_battery_capacity_khw,current_soc,'
          'electric_vehicle_departure_time,electric_vehicle_required_soc_departure,electric_vehicle_estimated_arrival_time,'
          'electric_vehicle_estimated_soc_arrival')


CURVE = [[0, 0.83], [0.3, 0.83], [0.7, 0.9], [0.8, 0.9], [1, 0.85]]       # the docstring example of Charger (electric_vehicle_charger.py:31-34)


def make(src: Path, dst: Path, seed: int, rows: int = 240, curves: bool = False) -> Path:
    """Copy the mini dataset at `src` to `dst` and overwrite its charger schedules.  Returns the schema path.
    `curves`: also give every other charger a charge efficiency curve and every third one a discharge curve."""
    import json
    if dst.exists():
        shutil.rmtree(dst)
    shutil.copytree(src, dst)
    schema = json.loads((dst / 'schema.json').read_text())
    evs = list(schema['electric_vehicles_def'].keys())
    caps = {k: v['battery']['attributes']['capacity'] for k, v in schema['electric_vehicles_def'].items()}
    chargers = [(b, c, cfg['charger_simulation']) for b, bs in schema['buildings'].items() for c, cfg in (bs.get('chargers') or {}).items()]
    rng = np.random.RandomState(seed)
    n = len(chargers)
    # timeline per EV: alternating away / (optional incoming) / connected segments; the charger it returns to rotates
    table = [[('3', '', '', '', '', '', '', '')] * rows for _ in range(n)]
    busy = np.zeros((rows, n), dtype=bool)
    for k, ev in enumerate(evs[:n]):
        t = 0 if rng.rand() < 0.5 else rng.randint(1, 6)
        visit = 0
        while t < rows:
            c = (k + visit * (1 if k % 2 == 0 else n - 1)) % n if rng.rand() < 0.35 else k      # sometimes another charger
            stay = rng.randint(2, 14)
            incoming = rng.randint(0, 3) if t > 0 else 0
            span = range(max(t - incoming, 0), min(t + stay, rows))
            if c == n - 1 or busy[list(span), c].any():          # the last charger is never used; no double booking
                t += rng.randint(1, 5)
                continue
            busy[list(span), c] = True
            announce = rng.rand() < 0.7                            # arrival SoC given or left empty
            soc_arr = round(float(rng.uniform(10, 95)), 1)
            for i, tt in enumerate(range(t - incoming, t)):
                if tt >= 0:
                    table[c][tt] = ('2', ev, caps[ev], round(float(rng.uniform(5, caps[ev])), 2), '', '', incoming - i - 1, soc_arr if announce else '')
            req = round(float(rng.uniform(50, 100)), 0)
            for i, tt in enumerate(range(t, min(t + stay, rows))):
                table[c][tt] = ('1', ev, caps[ev], round(float(rng.uniform(5, caps[ev])), 2), stay - i - 1, req, '', '')
            t += stay + rng.randint(1, 8)
            visit += 1
    for (b, c, fname), rows_ in zip(chargers, table):
        (dst / fname).write_text(HEADER + '\n' + '\n'.join(','.join(str(x) for x in r) for r in rows_) + '\n')
    if curves:
        for j, (b, c, _) in enumerate(chargers):
            attrs = schema['buildings'][b]['chargers'][c]['attributes']
            if j % 2 == 0:
                attrs['charge_efficiency_curve'] = CURVE
            if j % 3 == 0:
                attrs['discharge_efficiency_curve'] = [[0, 0.8], [0.5, 0.92], [1, 0.88]]
        (dst / 'schema.json').write_text(json.dumps(schema))
    return dst / 'schema.json'
