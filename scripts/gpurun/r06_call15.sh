#!/bin/bash
# Round 6, fifteenth GPU call: buildings per chunk of the thermal chain kernel (mode A, 1024 buildings) by batch size.
set -u
OUT=gpurun_out/r06o; mkdir -p $OUT; export TMPDIR=/tmp
for E in 1024 2048 4096 8192; do
  python bench.py --config C4 --precision chain --envs-per-gpu $E --reps 3 --steps 1000 > $OUT/C4_chain_${E}_default.json 2>/dev/null
  for bc in 32 64 128 256; do
    CL_TUNE_NW=16 CL_TUNE_B_CHUNK=$bc python bench.py --config C4 --precision chain --envs-per-gpu $E --reps 3 --steps 1000 > $OUT/C4_chain_${E}_bc$bc.json 2>$OUT/err.log
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06o/*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'value %.3e' % d['value'], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
    except Exception as e: print(f, 'unreadable', e)
PY
