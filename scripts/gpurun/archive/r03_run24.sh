#!/bin/bash
set -u
mkdir -p gpurun_out/r03_run24
timeout 900 python -m pytest tests/test_gpu_flex.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
timeout 300 python scripts/ev_step_bench.py > gpurun_out/r03_run24/ev_step_bench.log 2>&1; tail -1 gpurun_out/r03_run24/ev_step_bench.log
