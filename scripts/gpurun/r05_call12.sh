#!/bin/bash
# Round 5, twelfth GPU call: chunk size of the building-chunked thermal launch by batch size (BASELINE config 4 whole and its fractions).
set -u
OUT=gpurun_out/r05m; mkdir -p $OUT; export TMPDIR=/tmp
run() { # cfg E tag env...
  local cfg=$1 E=$2 tag=$3; shift 3
  env "$@" python bench.py --config $cfg --envs-per-gpu $E --no-cpu-baseline --steps 400 --warmup 40 > $OUT/${cfg}_${E}_$tag.json 2>$OUT/${cfg}_${E}_$tag.err || { echo "$cfg $E $tag FAILED: $(tail -1 $OUT/${cfg}_${E}_$tag.err | cut -c1-200)"; return; }
  python -c "
import json
d=json.load(open('$OUT/${cfg}_${E}_$tag.json')); r=d['roofline']
print('$cfg', $E, '$tag', 'launch_us %.2f'%r['launch_us'], 'frac %.3f'%r['frac'], r['kernel'])
"
}
for E in 8192 4096 2048 1024; do
  for bc in 32 64 128 256; do
    run C4 $E bc$bc CL_TUNE_B_CHUNK=$bc
  done
done
run C4 8192 bc64nt CL_TUNE_B_CHUNK=64 CL_TUNE_NT_STORES=1
run C4 8192 bc128nt CL_TUNE_B_CHUNK=128 CL_TUNE_NT_STORES=1
run C4 8192 bc64v1 CL_TUNE_B_CHUNK=64 CL_TUNE_VEC=1
run C4 8192 bc128v1 CL_TUNE_B_CHUNK=128 CL_TUNE_VEC=1
run C4-lean 8192 bc48 CL_TUNE_B_CHUNK=48
run C4-lean 8192 bc24 CL_TUNE_B_CHUNK=24
run C4-lean 8192 bc32v2 CL_TUNE_B_CHUNK=32 CL_TUNE_VEC=2
run C4-lean 8192 bc64v2 CL_TUNE_B_CHUNK=64 CL_TUNE_VEC=2
