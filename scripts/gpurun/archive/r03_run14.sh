#!/bin/bash
# both-demand models on the matrix-core kernel: the LSTM GPU tests, the generic-kernel timings, C3 re-timed, then the whole GPU suite
set -u
mkdir -p gpurun_out/r03_run14
timeout 900 python -m pytest tests/test_gpu_lstm.py -q -x 2>&1 | tail -8
timeout 300 python scripts/lstm_generic_bench.py > gpurun_out/r03_run14/lstm_generic_bench.log 2>&1; echo "generic bench rc=$?"; cat gpurun_out/r03_run14/lstm_generic_bench.log
timeout 300 python bench.py --config C3 > gpurun_out/r03_run14/bench_C3.json 2> gpurun_out/r03_run14/bench_C3.err; echo "C3 rc=$?"; cat gpurun_out/r03_run14/bench_C3.json
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
