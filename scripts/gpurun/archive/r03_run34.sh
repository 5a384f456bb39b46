#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_env_gpu.py tests/test_gpu_lstm.py -q -x 2>&1 | tail -4
