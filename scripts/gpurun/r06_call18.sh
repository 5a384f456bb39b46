#!/bin/bash
# Round 6, eighteenth GPU call: the fp32 C4 shard (1024 x 1024, mode A) at one env per lane -- waves per workgroup x buildings per chunk x LDS staging.
set -u
OUT=gpurun_out/r06r; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; env "$@" python bench.py --config C4 --precision fp32 --reps 3 > $OUT/$name.json 2>$OUT/$name.err; }
run default
for nw in 8 16; do for bc in 16 32 64; do
  run v1_nw${nw}_bc$bc CL_TUNE_VEC=1 CL_TUNE_NW=$nw CL_TUNE_B_CHUNK=$bc
  run v1_nolp_nw${nw}_bc$bc CL_TUNE_VEC=1 CL_TUNE_NW=$nw CL_TUNE_B_CHUNK=$bc CL_TUNE_FULL_VARIANT=3
done; done
run v2_nw8_bc16 CL_TUNE_VEC=2 CL_TUNE_NW=8 CL_TUNE_B_CHUNK=16
run v2_nw8_bc32 CL_TUNE_VEC=2 CL_TUNE_NW=8 CL_TUNE_B_CHUNK=32
run v2_nolp CL_TUNE_FULL_VARIANT=3
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06r/*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
    except Exception as e: print(f, 'unreadable', open(f.replace('.json','.err')).read()[-200:])
PY
