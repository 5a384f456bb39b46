"""Stage the REFERENCE package for the GPU box: `/root/reference` -> `oracle/_ref/reference/` (git-ignored, NOT gpurun-ignored).

TEST / MEASUREMENT INFRASTRUCTURE (oracle side).  `/root/reference` exists only in the build container; `bench.py`'s
`cpu_baseline` leg has to time the reference's own `CityLearnEnv.step` (citylearn.py:978-1056) "on the GPU box's own host cores
in the same run" (BASELINE.json north_star; SURVEY.md 8d last bullet).  The reference is pure Python -- there is nothing to
compile -- so its "build output" is a verbatim staged copy of exactly what that timing needs:

    citylearn/**            the package (sources, misc/settings.yaml, misc/queries/*.sql; no assets, no __pycache__)
    data/misc/battery_choices.yaml
    data/datasets/citylearn_challenge_2022_phase_all/      bench workload (17 buildings)
    data/datasets/citylearn_challenge_2022_phase_1/        C1 (5 buildings)

`oracle/_ref/` is listed in .gitignore: reference sources never enter the history; the directory travels to the GPU box with the
snapshot the same way the built `.so` files do.  `__graft_entry__.build()` runs this where `/root/reference` exists; on the GPU box
the prebuilt staging is used as is.  The product package (`citylearn_amd/`) never imports anything below `oracle/`.

    python oracle/ref_harness/stage_reference.py            # (re)stage; prints the manifest summary
"""
from __future__ import annotations

import hashlib
import json
import shutil
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
STAGED = HERE.parent / '_ref' / 'reference'
SOURCE = Path('/root/reference')
DATASETS = ('citylearn_challenge_2022_phase_all', 'citylearn_challenge_2022_phase_1')
_SKIP_DIRS = {'__pycache__', 'assets'}


def _copy_tree(src: Path, dst: Path, manifest: dict, rel_root: Path):
    for p in sorted(src.rglob('*')):
        if p.is_dir() or _SKIP_DIRS & set(p.relative_to(src).parts) or p.suffix == '.pyc':
            continue
        out = dst / p.relative_to(src)
        out.parent.mkdir(parents=True, exist_ok=True)
        shutil.copyfile(p, out)
        manifest[str(out.relative_to(rel_root))] = hashlib.sha256(out.read_bytes()).hexdigest()


def stage(source: Path = SOURCE, staged: Path = STAGED) -> dict:
    """Copy the pieces listed above and write MANIFEST.json (file -> sha256, version, when/where staged)."""
    if not (source / 'citylearn' / 'citylearn.py').is_file():
        raise RuntimeError(f'reference not found at {source}')
    if staged.exists():
        shutil.rmtree(staged)
    files: dict = {}
    _copy_tree(source / 'citylearn', staged / 'citylearn', files, staged)
    (staged / 'data' / 'misc').mkdir(parents=True, exist_ok=True)
    shutil.copyfile(source / 'data' / 'misc' / 'battery_choices.yaml', staged / 'data' / 'misc' / 'battery_choices.yaml')
    files['data/misc/battery_choices.yaml'] = hashlib.sha256((staged / 'data' / 'misc' / 'battery_choices.yaml').read_bytes()).hexdigest()
    for d in DATASETS:
        _copy_tree(source / 'data' / 'datasets' / d, staged / 'data' / 'datasets' / d, files, staged)
    import re
    m = re.search(r"__version__\s*=\s*['\"]([^'\"]+)['\"]", (staged / 'citylearn' / '__init__.py').read_text())      # (parsed, not executed)
    version = {'__version__': m.group(1) if m else None}
    manifest = {'what': 'verbatim staging of the reference package for the cpu_baseline timing (oracle/ref_harness/time_reference.py)',
                'source': str(source), 'version': version.get('__version__'), 'staged': time.strftime('%Y-%m-%d %H:%M:%S'),
                'n_files': len(files), 'bytes': sum((staged / f).stat().st_size for f in files), 'files': files}
    (staged / 'MANIFEST.json').write_text(json.dumps(manifest, indent=1) + '\n')
    return manifest


def verify(staged: Path = STAGED) -> bool:
    """True when every staged file still has the digest the manifest recorded (the timing refuses a tampered staging)."""
    try:
        manifest = json.loads((staged / 'MANIFEST.json').read_text())
    except (OSError, ValueError):
        return False
    return all((staged / f).is_file() and hashlib.sha256((staged / f).read_bytes()).hexdigest() == h
               for f, h in manifest['files'].items())


if __name__ == '__main__':
    m = stage()
    print(f"staged reference v{m['version']}: {m['n_files']} files, {m['bytes'] / 1e6:.1f} MB -> {STAGED}")
    sys.exit(0 if verify() else 1)
