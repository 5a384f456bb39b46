#!/bin/bash
# Round 5, twenty-fourth GPU call: selection thresholds re-measured after the load-hint change -- lean vs env-major kernel by batch size, envs per lane by batch size.
set -u
OUT=gpurun_out/r05y; mkdir -p $OUT; export TMPDIR=/tmp
run() { # E tag env...
  local E=$1 tag=$2; shift 2
  env "$@" python bench.py --envs-per-gpu $E --no-cpu-baseline --no-streaming --no-traffic-pass --no-chain-entry --steps 1500 --warmup 150 > $OUT/h_${E}_$tag.json 2>$OUT/h_${E}_$tag.err || { echo "$E $tag FAILED: $(tail -1 $OUT/h_${E}_$tag.err | cut -c1-160)"; return; }
  python -c "
import json
d=json.load(open('$OUT/h_${E}_$tag.json')); r=d['roofline']
print($E, '$tag', 'launch_us %.3f'%r['launch_us'], 'ns/kunit %.3f'%(r['launch_us']*1e6/(17*$E)), r['kernel'])
"
}
for E in 90112 98304 106496 114688 131072 163840 196608 262144; do
  run $E default
  run $E lean CL_TUNE_LEAN_VARIANT=2 CL_TUNE_ENVMAJOR=2
  run $E envmajor CL_TUNE_ENVMAJOR=1
done
for E in 20480 24576 32768 40960 49152; do
  run $E default
  run $E vec4 CL_TUNE_VEC=4
  run $E vec2 CL_TUNE_VEC=2
done
for E in 8192 12288 16384; do
  run $E default
  run $E vec1 CL_TUNE_VEC=1
  run $E vec2 CL_TUNE_VEC=2
done
