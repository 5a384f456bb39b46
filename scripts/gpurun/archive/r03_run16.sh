#!/bin/bash
# capture_rollout: test + user-level bench
set -u
mkdir -p gpurun_out/r03_run16
timeout 600 python -m pytest tests/test_env_gpu.py -q -x -k "captured" 2>&1 | tail -15
timeout 900 python scripts/env_step_bench.py > gpurun_out/r03_run16/env_step_bench.log 2>&1; echo "bench rc=$?"; cat gpurun_out/r03_run16/env_step_bench.log | tail -20
