#!/bin/bash
# Round 6, ninth GPU call: the packed thermal rollout with its Philox blocks cached in LDS -- tests, BASELINE config 4 in mode B on both precision
# models, the 9-building district in mode B, and the SQ counters of the new kernel (vector instructions per unit-step for its VALU roofline).
set -u
OUT=gpurun_out/r06i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rollout.py -m gpu -q > $OUT/rollout_tests.log 2>&1
echo "rollout tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/rollout_tests.log | tail -20
for p in chain fp32; do
  for E in 1024 8192; do
    python bench.py --config C4-B --envs-per-gpu $E --precision $p > $OUT/c4b_${p}_$E.json 2>$OUT/c4b_${p}_$E.err
  done
done
CTR="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
for p in chain fp32; do
  rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc_c4b_$p -o run -- python bench.py --config C4-B --precision $p --steps 12 --warmup 3 --reps 1 --no-graph > /dev/null 2>$OUT/pmc_c4b_$p.log
  python scripts/pmc_by_kernel.py cl_rollout $OUT/pmc_c4b_$p/*counter_collection.csv > $OUT/r06_c4b_${p}_sq_by_kernel.jsonl 2>>$OUT/pmc_c4b_$p.log
  grep cl_rollout_full $OUT/r06_c4b_${p}_sq_by_kernel.jsonl
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06i/*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'value %.3e' % d['value'], 'ms_per_step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], 'frac', r['frac'], r['kernel'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
