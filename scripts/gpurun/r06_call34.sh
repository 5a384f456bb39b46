#!/bin/bash
set -u
OUT=gpurun_out/r06z4; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rollout.py -m gpu -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -20
for p in chain fp32; do
  python bench.py --config T9 --precision $p --reps 3 > $OUT/T9_$p.json 2>/dev/null
  python bench.py --config C4-B --precision $p --reps 3 > $OUT/C4-B_$p.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06z4/*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'launch_us %.2f' % r['launch_us'], r['kernel'], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.get('mode_b', {}).items() if k != 'what'})
PY
