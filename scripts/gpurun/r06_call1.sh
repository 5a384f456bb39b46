#!/bin/bash
# Round 6, first GPU call: (1) the parity table at the north-star bar with NO slack factors -- every fp32 / chain / f64 parity test in measuring
# mode (CL_PARITY_MEASURE: record, do not fail), the new full-year free-running test included; (2) the rest of the GPU suite as it stands
# (ticket words moved to the plane's tail, chain branch by the clamped energy's sign); (3) the SIMD-balance probe of the headline launch.
set -u
OUT=gpurun_out/r06a; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/parity.jsonl
CL_PARITY_REPORT=$OUT/parity.jsonl CL_PARITY_MEASURE=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_sizes.py -m gpu -q -s > $OUT/parity_measure.log 2>&1
echo "parity measure rc=$?"; tail -3 $OUT/parity_measure.log
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_config_sizes.py > $OUT/suite_rest.log 2>&1
echo "rest of suite rc=$?"; tail -3 $OUT/suite_rest.log
timeout 600 python scripts/r06_balance_probe.py 65536 > $OUT/balance_65536.log 2>&1; echo "balance rc=$?"; cat $OUT/balance_65536.log
