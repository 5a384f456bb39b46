#!/bin/bash
# streaming KPIs inside the thermal step launch: tests + cost
set -u
mkdir -p gpurun_out/r03_run17
timeout 900 python -m pytest tests/test_gpu_config_sizes.py tests/test_env_gpu.py tests/test_gpu_bench.py -q -x 2>&1 | tail -8
timeout 300 python scripts/kpi_cost_probe.py > gpurun_out/r03_run17/kpi_cost_probe.log 2>&1; echo "probe rc=$?"; cat gpurun_out/r03_run17/kpi_cost_probe.log
