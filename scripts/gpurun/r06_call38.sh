#!/bin/bash
# Round 6, thirty-eighth GPU call: the LSTM kernel capped at 168 registers (three waves per SIMD, 312 bytes of scratch per lane) against the 245-register default.
set -u
OUT=gpurun_out/r06z7; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py --config C3 --reps 3 > $OUT/C3_default.json 2>/dev/null
CITYLEARN_AMD_LIB=citylearn_amd/libcitylearn_amd_lstm3.so python bench.py --config C3 --reps 3 > $OUT/C3_wpe3.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06z7/*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
PY
