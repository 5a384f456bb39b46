#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lstm.py -q -x -k "sweep or reference_cooling or cell_update" 2>&1 | tail -4
