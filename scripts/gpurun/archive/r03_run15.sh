#!/bin/bash
# CLD_LSTM_TWO_DEMANDS: the LSTM GPU tests, the stage timings, C3 re-timed
set -u
mkdir -p gpurun_out/r03_run15
timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_bench.py -q -x 2>&1 | tail -8
timeout 300 python scripts/lstm_generic_bench.py > gpurun_out/r03_run15/lstm_generic_bench.log 2>&1; echo "generic bench rc=$?"; cat gpurun_out/r03_run15/lstm_generic_bench.log
timeout 300 python bench.py --config C3 > gpurun_out/r03_run15/bench_C3.json 2> gpurun_out/r03_run15/bench_C3.err; echo "C3 rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r03_run15/bench_C3.json')); print(d['ms_per_step'], d['roofline']['launch_us'], d['roofline']['kernel'], d['roofline']['frac'])"
