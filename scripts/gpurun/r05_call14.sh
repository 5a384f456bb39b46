#!/bin/bash
# Round 5, fourteenth GPU call: A/B of the generalised deferred fold against the previous build (CITYLEARN_AMD_LIB) on the shapes whose
# geometry did not change (1024 x 1024 / 2048 shards), alternating on one box; then the changed shapes once more.
set -u
OUT=gpurun_out/r05o; mkdir -p $OUT; export TMPDIR=/tmp
PREV=$PWD/citylearn_amd/libcl_prev.so
run() { # cfg E tag env...
  local cfg=$1 E=$2 tag=$3; shift 3
  env "$@" python bench.py --config $cfg --envs-per-gpu $E --no-cpu-baseline --steps 600 --warmup 60 > $OUT/${cfg}_${E}_$tag.json 2>$OUT/${cfg}_${E}_$tag.err || { echo "$cfg $E $tag FAILED: $(tail -1 $OUT/${cfg}_${E}_$tag.err | cut -c1-200)"; return; }
  python -c "
import json
d=json.load(open('$OUT/${cfg}_${E}_$tag.json')); r=d['roofline']
print('$cfg', $E, '$tag', 'launch_us %.2f'%r['launch_us'], 'frac %.3f'%r['frac'], r['kernel'])
"
}
for rep in a b c; do
  for E in 1024 2048; do
    run C4 $E new_$rep
    run C4 $E prev_$rep CITYLEARN_AMD_LIB=$PREV
  done
  run C4-lean 1024 new_$rep
  run C4-lean 1024 prev_$rep CITYLEARN_AMD_LIB=$PREV
done
run C4 8192 new_a; run C4 8192 prev_a CITYLEARN_AMD_LIB=$PREV
run C4-lean 8192 new_a; run C4-lean 8192 prev_a CITYLEARN_AMD_LIB=$PREV
run C4 2048 bc64_a CL_TUNE_B_CHUNK=64
run C4 2048 bc48_a CL_TUNE_B_CHUNK=48
