#!/bin/bash
set -u
O=gpurun_out/r04_run9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_sizes.py -x -q 2>&1 | tail -3
timeout 600 python scripts/tp_pipeline_sweep.py > $O/tp_pipeline_sweep.log 2>&1; cat $O/tp_pipeline_sweep.log
