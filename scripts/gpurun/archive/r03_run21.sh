#!/bin/bash
set -u
mkdir -p gpurun_out/r03_run21
timeout 900 python -m pytest tests/test_gpu_config_sizes.py tests/test_env_gpu.py tests/test_gpu_offsets.py tests/test_gpu_rollout.py -q -x 2>&1 | tail -5
timeout 300 python scripts/kpi_cost_probe.py > gpurun_out/r03_run21/kpi_in_step_probe.log 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids gpurun_out/r03_run21/kpi_in_step_probe.log
