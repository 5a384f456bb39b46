#!/bin/bash
set -u
timeout 1200 python -m pytest tests/test_gpu_bench.py -q -x 2>&1 | tail -5
