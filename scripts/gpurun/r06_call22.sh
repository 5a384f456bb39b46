#!/bin/bash
set -u
OUT=gpurun_out/r06v; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rollout.py -m gpu -q > $OUT/rollout_tests.log 2>&1
echo "rollout tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/rollout_tests.log | tail -20
for p in chain fp32; do
  python bench.py --config C4-B --precision $p --reps 3 > $OUT/C4-B_${p}_1024.json 2>/dev/null
  python bench.py --config C4-B --precision $p --envs-per-gpu 8192 --reps 3 > $OUT/C4-B_${p}_8192.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06v/*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], 'frac', r['frac'], r['kernel'])
PY
