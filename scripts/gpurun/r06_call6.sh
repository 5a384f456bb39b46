#!/bin/bash
# Round 6, sixth GPU call: thermal step + observe in one launch (tests + timings), envs per lane of the chain's chunk kernel on the C4-lean shard,
# the pinned selection map, the parity record of the final tree.
set -u
OUT=gpurun_out/r06f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_observe.py -m gpu -q > $OUT/observe_tests.log 2>&1; echo "observe tests rc=$?"; tail -3 $OUT/observe_tests.log
timeout 600 python scripts/step_observe_bench.py > $OUT/step_observe_bench.log 2>&1; echo "sob rc=$?"; cat $OUT/step_observe_bench.log
CL_SOB_FP32=1 timeout 600 python scripts/step_observe_bench.py > $OUT/step_observe_bench_fp32.log 2>&1; cat $OUT/step_observe_bench_fp32.log
for cfg in C4-lean; do for E in 1024 8192; do for v in 1 2 4; do
  CL_TUNE_VEC=$v python bench.py --config $cfg --envs-per-gpu $E --steps 1000 --reps 3 > $OUT/c4lean_${E}_vec$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/c4lean_${E}_vec$v.json')); r=d['roofline']; print('$cfg', $E, 'vec', $v, 'launch_us %.2f'%r['launch_us'], 'frac %.3f'%r['frac'], r['kernel'])"
done; done; done
rm -f $OUT/parity.jsonl
CL_PARITY_REPORT=$OUT/parity.jsonl timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_flex.py tests/test_gpu_check.py tests/test_gpu_checkpoint.py -m gpu -q > $OUT/parity_suite.log 2>&1
echo "parity suite rc=$?"; tail -3 $OUT/parity_suite.log
