#!/bin/bash
# tp kernel with next-item prefetch: parity + timing
set -u
mkdir -p gpurun_out/r03_run32
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_sizes.py -q -x 2>&1 | tail -3
for rep in 1 2; do timeout 200 python scripts/alt_lib_time.py thermal thermal256k p3 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03_run32/tp_prefetch.log
