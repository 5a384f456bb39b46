#!/bin/bash
# Round 6, second GPU call: the whole GPU suite under the new defaults (CLD_F64_CHAIN where supported, CLD_CHECK in CityLearnEnv), with the new
# checkpoint / CLD_CHECK tests; failures listed, not stopped at.
set -u
OUT=gpurun_out/r06b; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/parity.jsonl
CL_PARITY_REPORT=$OUT/parity.jsonl timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench.py -x --maxfail=40 > $OUT/suite.log 2>&1
echo "suite rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/suite.log | tail -50
