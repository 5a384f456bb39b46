#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_gpu_lstm.py -q -x -k "mid_episode or comfort_kpis" 2>&1 | tail -15
