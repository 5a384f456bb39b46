"""The golden recipe is re-runnable: where /root/reference exists (the build container), `oracle/ref_harness/gen_golden.py`
regenerates a PACKAGE-DATA fixture (the three headline ones live next to the package, not under tests/golden/<name>/dataset) into
a scratch directory and the result equals the committed files bit for bit -- the mini dataset and every array of reference.npz.
Also: the staging of the reference for bench.py's cpu_baseline leg (oracle/_ref/) matches its manifest, and is never imported by
the product package."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
HARNESS = ROOT / 'oracle' / 'ref_harness'
sys.path.insert(0, str(HARNESS))

needs_reference = pytest.mark.skipif(not Path('/root/reference/citylearn/citylearn.py').is_file(),
                                     reason='/root/reference exists only in the build container')


@needs_reference
def test_package_data_fixture_regenerates_bit_for_bit(tmp_path):
    """g2023_p2 (3 buildings x 719 steps: outage path, LSTM, ComfortReward) is the quickest of the three package-data fixtures."""
    name = 'g2023_p2'
    env = {**os.environ, 'CL_GOLDEN_ROOT': str(tmp_path)}
    p = subprocess.run([sys.executable, str(HARNESS / 'gen_golden.py'), '--one', 'reference', name], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    new, old = np.load(tmp_path / name / 'reference.npz'), np.load(ROOT / 'tests' / 'golden' / name / 'reference.npz')
    assert sorted(new.files) == sorted(old.files)
    for k in new.files:
        a, b = new[k], old[k]
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert np.array_equal(a, b, equal_nan=a.dtype.kind == 'f'), k
    import gen_golden
    committed = gen_golden.PACKAGE_DATASETS[name]
    files = sorted(q.relative_to(committed) for q in committed.rglob('*') if q.is_file())
    assert files == sorted(q.relative_to(tmp_path / name / 'dataset') for q in (tmp_path / name / 'dataset').rglob('*') if q.is_file())
    for f in files:
        assert (committed / f).read_bytes() == (tmp_path / name / 'dataset' / f).read_bytes(), f


def test_recipe_does_not_delete_sibling_fixture_files():
    """`run_reference` used to rmtree tests/golden/<name>/ -- observations.npz, kpi_conditions.json and kpi_mid.npz with it -- and then
    write into the directory it had just removed (FileNotFoundError for the package-data fixtures)."""
    src = (HARNESS / 'gen_golden.py').read_text()
    body = src[src.index('def run_reference('):src.index('def run_observations(')]
    assert 'shutil.rmtree(out_dir)' not in body and 'out_dir.mkdir(parents=True, exist_ok=True)' in body


@needs_reference
def test_reference_staging_matches_its_manifest(tmp_path):
    import stage_reference
    m = stage_reference.stage(staged=tmp_path / 'reference')
    assert m['version'] == '2.4.2' and stage_reference.verify(tmp_path / 'reference')
    assert 'citylearn/citylearn.py' in m['files'] and 'data/datasets/citylearn_challenge_2022_phase_all/schema.json' in m['files']
    assert not [f for f in m['files'] if '__pycache__' in f or f.endswith('.pyc') or f.startswith('citylearn/assets')]
    (tmp_path / 'reference' / 'citylearn' / 'citylearn.py').write_text('tampered')
    assert not stage_reference.verify(tmp_path / 'reference')


def test_staging_is_git_ignored_and_unreachable_from_the_product():
    assert 'oracle/_ref/' in (ROOT / '.gitignore').read_text().split()
    if (ROOT / '.gpurunignore').exists():
        assert 'oracle/_ref' not in (ROOT / '.gpurunignore').read_text()
    tracked = subprocess.run(['git', 'ls-files', 'oracle/_ref'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    assert tracked == ''
    for py in (ROOT / 'citylearn_amd').rglob('*.py'):
        text = py.read_text()
        assert '_ref' not in text or 'oracle' not in text, py
        assert 'ref_harness' not in text and 'import oracle' not in text and 'from oracle' not in text, py


def test_bench_reference_leg_reports_the_fallback_honestly(tmp_path, monkeypatch):
    """Without a staging the line must not pretend: the committed build-container timing is attached and labelled as such."""
    sys.path.insert(0, str(ROOT))
    import bench
    monkeypatch.setattr(bench, 'ROOT', tmp_path)
    (tmp_path / 'profiles').mkdir()
    (tmp_path / 'profiles' / 'reference_cpu_timing.json').write_text(json.dumps({'kind': 'reference', 'value': 1.0, 'host': 'somewhere else'}))
    ref = bench.reference_cpu_baseline(2)
    assert ref['measured'].startswith('NOT in this run') and 'not staged' in ref['live_error'] and ref['value'] == 1.0
