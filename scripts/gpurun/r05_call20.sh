#!/bin/bash
# Round 5, twentieth GPU call: the latency-ordered lean kernel's non-temporal LOADS, A/B builds alternating (headline 17 x 65 536, C2, 17 x 32 768 / 16 384).
set -u
OUT=gpurun_out/r05u; mkdir -p $OUT; export TMPDIR=/tmp
ALT=$PWD/citylearn_amd/libcl_plainloads.so
run() { # E tag env...
  local E=$1 tag=$2; shift 2
  env "$@" python bench.py --envs-per-gpu $E --no-cpu-baseline --no-streaming --no-traffic-pass --no-chain-entry --steps 2000 --warmup 200 > $OUT/h_${E}_$tag.json 2>$OUT/h_${E}_$tag.err || { echo "$E $tag FAILED"; return; }
  python -c "
import json
d=json.load(open('$OUT/h_${E}_$tag.json')); r=d['roofline']
print('headline', $E, '$tag', 'launch_us %.3f'%r['launch_us'], 'frac %.3f'%r['frac'], r['kernel'])
"
}
for rep in a b c; do
  for E in 65536 32768; do
    run $E nt_$rep
    run $E plain_$rep CITYLEARN_AMD_LIB=$ALT
  done
done
run 16384 nt_a; run 16384 plain_a CITYLEARN_AMD_LIB=$ALT
run 81920 nt_a; run 81920 plain_a CITYLEARN_AMD_LIB=$ALT
