import os
import sys
from pathlib import Path

# Two OpenMP pools end up in one test process: libgomp under oracle/libcl_oracle.so (the C port of the oracle) and the one torch brings.
# With the default wait policy the idle workers of one pool SPIN while the other pool computes -- on an 8-core box that turned single
# tests from seconds into minutes (a 3.5-minute suite once took 23 minutes).  Idle workers sleep instead; set before either runtime loads.
os.environ.setdefault('OMP_WAIT_POLICY', 'passive')
os.environ.setdefault('GOMP_SPINCOUNT', '0')

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
