#!/bin/bash
set -u
OUT=gpurun_out/r06z2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_config_sizes.py -m gpu -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -20
for E in 1024 8192; do python bench.py --config C4-lean-B --envs-per-gpu $E --reps 3 > $OUT/C4leanB_chain_$E.json 2>/dev/null; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06z2/*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
PY
