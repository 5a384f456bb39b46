#!/bin/bash
# Round 6, sixteenth GPU call: chunk-size sweep of the chunked thermal kernels (128 .. 1024 buildings), both precision models; the config-size and rollout tests.
set -u
OUT=gpurun_out/r06p; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/chunks_*.jsonl
timeout 600 python scripts/r06_chunk_sweep.py $OUT/chunks_chain.jsonl chain 2>$OUT/sweep_chain.err | tail -20
timeout 600 python scripts/r06_chunk_sweep.py $OUT/chunks_fp32.jsonl fp32 2>$OUT/sweep_fp32.err | tail -20
timeout 900 python -m pytest tests/test_gpu_config_sizes.py tests/test_gpu_rollout.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail
