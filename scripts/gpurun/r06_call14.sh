#!/bin/bash
# Round 6, fourteenth GPU call: BASELINE config 4's thermal shard in mode A under the float64 chain -- eight-wave workgroups (80 registers, three per CU)
# against the sixteen-wave default, by buildings per chunk.
set -u
OUT=gpurun_out/r06n; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py --config C4 --precision chain --reps 3 > $OUT/C4_chain_default.json 2>/dev/null
for bc in 8 16 24 32 48; do
  CL_TUNE_NW=8 CL_TUNE_B_CHUNK=$bc python bench.py --config C4 --precision chain --reps 3 > $OUT/C4_chain_nw8_bc$bc.json 2>$OUT/err_$bc.log
done
for bc in 16 48 64; do
  CL_TUNE_NW=16 CL_TUNE_B_CHUNK=$bc python bench.py --config C4 --precision chain --reps 3 > $OUT/C4_chain_nw16_bc$bc.json 2>$OUT/err16_$bc.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06n/*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'value %.3e' % d['value'], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
    except Exception as e: print(f, 'unreadable', e)
PY
