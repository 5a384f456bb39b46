#!/bin/bash
set -u
O=gpurun_out/r04_run13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_observe.py -x -q 2>&1 | tail -3
timeout 600 python scripts/observe_bench.py > $O/observe_bench.log 2>$O/observe_bench.err
grep -v "compact form\|all-exogenous" $O/observe_bench.log; tail -3 $O/observe_bench.err
