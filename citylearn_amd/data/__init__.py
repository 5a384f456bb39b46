"""Sample datasets shipped with the package (see README.md in this directory)."""
from pathlib import Path

_HERE = Path(__file__).resolve().parent


def sample_schema(name: str = 'citylearn_challenge_2022_phase_all_720h') -> str:
    """Path of the `schema.json` of a shipped sample dataset."""
    path = _HERE / name / 'schema.json'
    if not path.exists():
        raise FileNotFoundError(f'no sample dataset {name!r} under {_HERE}')
    return str(path)
