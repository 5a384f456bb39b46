#!/bin/bash
set -u
OUT=gpurun_out/r06z; mkdir -p $OUT; export TMPDIR=/tmp
for E in 1024 8192; do
  python bench.py --config C4-lean-B --precision chain --envs-per-gpu $E --reps 3 > $OUT/C4leanB_chain_${E}_default.json 2>/dev/null
  CL_TUNE_VEC=2 python bench.py --config C4-lean-B --precision chain --envs-per-gpu $E --reps 3 > $OUT/C4leanB_chain_${E}_vec2.json 2>$OUT/err_$E.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06z/*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
    except Exception as e: print(f, 'unreadable', open(f.replace('.json','.log').replace('C4leanB_chain_','err_').replace('_vec2','')).read()[-300:] if 'vec2' in f else e)
PY
