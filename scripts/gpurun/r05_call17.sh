#!/bin/bash
# Round 5, seventeenth GPU call: workgroup size of the chunked thermal launch (16 waves x 1 workgroup per CU against 8 waves x 2), second launch
# everywhere (finish = 1) so that only the step kernel differs.
set -u
OUT=gpurun_out/r05r; mkdir -p $OUT; export TMPDIR=/tmp
run() { # cfg E tag env...
  local cfg=$1 E=$2 tag=$3; shift 3
  env "$@" python bench.py --config $cfg --envs-per-gpu $E --no-cpu-baseline --steps 400 --warmup 40 > $OUT/${cfg}_${E}_$tag.json 2>$OUT/${cfg}_${E}_$tag.err || { echo "$cfg $E $tag FAILED: $(tail -1 $OUT/${cfg}_${E}_$tag.err | cut -c1-200)"; return; }
  python -c "
import json
d=json.load(open('$OUT/${cfg}_${E}_$tag.json')); r=d['roofline']
print('$cfg', $E, '$tag', 'launch_us %.2f'%r['launch_us'], 'frac %.3f'%r['frac'], r['kernel'])
"
}
for rep in a b; do
  run C4 8192 bc128nw16_$rep CL_TUNE_FINISH=1 CL_TUNE_B_CHUNK=128 CL_TUNE_NW=16
  run C4 8192 bc64nw8_$rep CL_TUNE_FINISH=1 CL_TUNE_B_CHUNK=64 CL_TUNE_NW=8
  run C4 8192 bc128nw8_$rep CL_TUNE_FINISH=1 CL_TUNE_B_CHUNK=128 CL_TUNE_NW=8
  run C4 8192 bc32nw4_$rep CL_TUNE_FINISH=1 CL_TUNE_B_CHUNK=32 CL_TUNE_NW=4
  run C4 1024 bc32nw16_$rep CL_TUNE_FINISH=1 CL_TUNE_B_CHUNK=32 CL_TUNE_NW=16
  run C4 1024 bc16nw8_$rep CL_TUNE_FINISH=1 CL_TUNE_B_CHUNK=16 CL_TUNE_NW=8
  run C4 1024 bc32nw8_$rep CL_TUNE_FINISH=1 CL_TUNE_B_CHUNK=32 CL_TUNE_NW=8
  run C4 1024 bc64nw16_$rep CL_TUNE_FINISH=1 CL_TUNE_B_CHUNK=64 CL_TUNE_NW=16
done
