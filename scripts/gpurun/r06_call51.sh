#!/bin/bash
# Round 6, fifty-first GPU call: the parity record of the final tree (every parity check of the GPU suite, worst error in units of the bar).
set -u
OUT=gpurun_out/r06z9; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/parity.jsonl
CL_PARITY_REPORT=$OUT/parity.jsonl timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_flex.py tests/test_gpu_check.py tests/test_gpu_checkpoint.py tests/test_env_gpu.py tests/test_gpu_observe.py -m gpu -q > $OUT/parity_suite.log 2>&1
echo "parity suite rc=$?"; tail -3 $OUT/parity_suite.log
python scripts/parity_table.py $OUT/parity.jsonl > $OUT/parity_worst.md; head -6 $OUT/parity_worst.md
