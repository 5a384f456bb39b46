#!/bin/bash
set -u
OUT=gpurun_out/r06z3; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_offsets.py -m gpu -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -20
for p in chain fp32; do
  python bench.py --config C5 --precision $p --reps 5 > $OUT/C5_$p.json 2>/dev/null
  python bench.py --config C4-lean-B --precision $p --reps 3 > $OUT/C4leanB_${p}_1024.json 2>/dev/null
  python bench.py --config C4-lean-B --precision $p --envs-per-gpu 8192 --reps 3 > $OUT/C4leanB_${p}_8192.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06z3/*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
PY
