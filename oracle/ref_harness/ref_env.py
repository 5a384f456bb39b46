"""Import helper for running the *reference* CityLearn (read-only at /root/reference).

TEST INFRASTRUCTURE ONLY (oracle side).  Used by `gen_golden.py` to produce the
committed fixtures under tests/golden/ and by `time_reference.py` (bench.py's
`cpu_baseline.reference` leg, a subprocess).  `/root/reference` does not exist on the
GPU box: there the root is the git-ignored staging `oracle/_ref/reference`
(`stage_reference.py`); nothing on the `-m gpu` / smoke path imports this module.

Recipe: SURVEY.md App. C — gymnasium + simplejson stand-ins, a pre-seeded
platformdirs cache so `CityLearnEnv._load` (citylearn.py:2055-2057) never hits
the network, and schemas passed as file paths (citylearn.py:863-883).
"""
import os
import shutil
import sys
import tempfile
from pathlib import Path

_HERE = Path(__file__).resolve().parent
STAGED_ROOT = _HERE.parent / '_ref' / 'reference'        # stage_reference.py: what the GPU box has instead of /root/reference


def _default_root() -> Path:
    if 'CITYLEARN_REFERENCE_ROOT' in os.environ:
        return Path(os.environ['CITYLEARN_REFERENCE_ROOT'])
    live = Path('/root/reference')
    return live if (live / 'citylearn' / 'citylearn.py').is_file() else STAGED_ROOT


REFERENCE_ROOT = _default_root()


def reference_available() -> bool:
    return (REFERENCE_ROOT / 'citylearn' / 'citylearn.py').is_file()


def setup_reference():
    """Put the reference + stubs on sys.path and seed the dataset cache. Returns the citylearn module."""
    if not reference_available():
        raise RuntimeError(f'reference not found at {REFERENCE_ROOT}')
    xdg = Path(tempfile.gettempdir()) / 'citylearn_ref_xdg'
    misc = xdg / 'citylearn' / 'v2.4.2' / 'misc'
    misc.mkdir(parents=True, exist_ok=True)
    shutil.copy(REFERENCE_ROOT / 'data' / 'misc' / 'battery_choices.yaml', misc / 'battery_choices.yaml')
    pv = misc / 'lbl-tracking_the_sun-res-pv.csv'
    if not pv.exists():
        pv.write_text('nameplate_capacity_module_1,inverter_loading_ratio,tilt_1,azimuth_1,'
                      'bifacial_module_1,PV_system_size_DC,module_area\n')
    os.environ['XDG_CACHE_HOME'] = str(xdg)
    for p in (str(_HERE / 'stubs'), str(REFERENCE_ROOT)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import citylearn  # noqa: E402
    assert Path(citylearn.__file__).resolve().is_relative_to(REFERENCE_ROOT), citylearn.__file__
    return citylearn


def dataset_schema(name: str) -> str:
    return str(REFERENCE_ROOT / 'data' / 'datasets' / name / 'schema.json')
