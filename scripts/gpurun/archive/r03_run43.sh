#!/bin/bash
set -u
timeout 1200 python -m pytest tests/test_env_gpu.py tests/test_gpu_lstm.py tests/test_gpu_flex.py tests/test_gym_surface.py -q -x 2>&1 | tail -6
