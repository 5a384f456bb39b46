#!/bin/bash
set -u
OUT=gpurun_out/r06u; mkdir -p $OUT; export TMPDIR=/tmp
for p in chain fp32; do
  python bench.py --config C5 --precision $p --reps 3 > $OUT/C5_${p}_default.json 2>/dev/null
  CL_TUNE_VEC=1 python bench.py --config C5 --precision $p --reps 3 > $OUT/C5_${p}_vec1.json 2>/dev/null
  for E in 16384 65536; do
    python bench.py --config C5 --precision $p --envs-per-gpu $E --reps 3 > $OUT/C5_${p}_${E}_default.json 2>/dev/null
    CL_TUNE_VEC=1 python bench.py --config C5 --precision $p --envs-per-gpu $E --reps 3 > $OUT/C5_${p}_${E}_vec1.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06u/*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
PY
