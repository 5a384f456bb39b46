#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_config_sizes.py tests/test_gpu_bench.py -x -q 2>&1 | tail -3
