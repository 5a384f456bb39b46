#!/bin/bash
# round-4 profile recipe + the GPU suite on the same tree
set -u
O=gpurun_out/r04_run10; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?" >> $O/gpu_suite.log; tail -3 $O/gpu_suite.log
timeout 2400 bash scripts/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -40 $O/profile_round.log
