#!/bin/bash
set -u
OUT=gpurun_out/r03_run8
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_env_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout 600 python scripts/lstm_generic_bench.py > $OUT/lstm_generic_bench.log 2>&1; cat $OUT/lstm_generic_bench.log
