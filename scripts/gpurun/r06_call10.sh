#!/bin/bash
# Round 6, tenth GPU call: A/B of the packed thermal rollout's parameter reads -- re-read per step (default), header words hoisted (1), everything hoisted (2).
set -u
OUT=gpurun_out/r06j; mkdir -p $OUT; export TMPDIR=/tmp
for v in default hoist1 hoist2; do
  lib=citylearn_amd/libcitylearn_amd.so; [ $v != default ] && lib=citylearn_amd/libcitylearn_amd_$v.so
  for p in chain fp32; do
    for c in C4-B; do
      CITYLEARN_AMD_LIB=$lib python bench.py --config $c --precision $p --reps 3 > $OUT/${c}_${p}_$v.json 2>/dev/null
    done
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06j/*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
