#!/bin/bash
set -u
mkdir -p gpurun_out/r03_run19
timeout 300 python scripts/kpi_cost_probe.py > gpurun_out/r03_run19/kv_nt_probe.log 2>&1; grep -v amdgpu.ids gpurun_out/r03_run19/kv_nt_probe.log
