#!/bin/bash
# Round 6, eighth GPU call: the packed thermal rollout kernel -- its tests, then BASELINE config 4 in mode B on both
# precision models beside mode A, and the C2 line with its mode-B side entry.
set -u
OUT=gpurun_out/r06h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_observe.py -m gpu -q > $OUT/rollout_tests.log 2>&1
echo "rollout tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/rollout_tests.log | tail -20
for p in chain fp32; do
  for E in 1024 8192; do
    python bench.py --config C4-B --envs-per-gpu $E --precision $p > $OUT/c4b_${p}_$E.json 2>$OUT/c4b_${p}_$E.err
    python bench.py --config C4 --envs-per-gpu $E --precision $p --reps 1 > $OUT/c4_${p}_$E.json 2>/dev/null
  done
  CL_TUNE_FULL_VARIANT=1 python bench.py --config C4-B --envs-per-gpu 1024 --precision $p > $OUT/c4b_${p}_1024_scalar.json 2>/dev/null
done
python bench.py --config C2 > $OUT/c2.json 2>$OUT/c2.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06h/*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'value %.3e' % d['value'], 'ms_per_step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], 'frac', r['frac'], r['kernel'], r.get('mode_b'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
timeout 600 python -m pytest tests/test_gpu_bench.py -m gpu -q > $OUT/bench_tests.log 2>&1
echo "bench tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/bench_tests.log | tail
