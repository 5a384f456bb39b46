#!/bin/bash
# Round 6, twelfth GPU call: the packed thermal rollout after the refresh loop walks active slots only -- tests, timings, counters, and smaller workgroups.
set -u
OUT=gpurun_out/r06l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rollout.py -m gpu -q > $OUT/rollout_tests.log 2>&1
echo "rollout tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/rollout_tests.log | tail -20
CTR1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for p in chain fp32; do
  python bench.py --config C4-B --precision $p --reps 3 > $OUT/C4-B_${p}_default.json 2>/dev/null
  for nw in 8 10 12; do
    CL_TUNE_B_CHUNK=$nw CL_TUNE_NW=$nw python bench.py --config C4-B --precision $p --reps 3 > $OUT/C4-B_${p}_nw$nw.json 2>$OUT/C4-B_${p}_nw$nw.err
  done
  python bench.py --config C4-B --precision $p --envs-per-gpu 8192 --reps 3 > $OUT/C4-B_${p}_8192.json 2>/dev/null
  rocprofv3 --pmc $CTR1 --output-format csv -d $OUT/pmc_${p} -o run -- python bench.py --config C4-B --precision $p --steps 12 --warmup 3 --reps 1 --no-graph > /dev/null 2>$OUT/pmc_${p}.log
  python scripts/pmc_by_kernel.py cl_rollout $OUT/pmc_${p}/*counter_collection.csv > $OUT/sq_${p}.jsonl; cat $OUT/sq_${p}.jsonl
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06l/*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
    except Exception as e: print(f, 'unreadable', e)
PY
