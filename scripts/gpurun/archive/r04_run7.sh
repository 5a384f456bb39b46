#!/bin/bash
set -u
O=gpurun_out/r04_run7; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?" >> $O/gpu_suite.log
tail -25 $O/gpu_suite.log
