#!/bin/bash
# Round 5, eleventh GPU call: BASELINE config 4 whole on one GPU (1024 x 8192) -- non-temporal stores and chunk geometry, alternating.
set -u
OUT=gpurun_out/r05l; mkdir -p $OUT; export TMPDIR=/tmp
run() { # cfg E tag env...
  local cfg=$1 E=$2 tag=$3; shift 3
  env "$@" python bench.py --config $cfg --envs-per-gpu $E --no-cpu-baseline --steps 600 --warmup 60 > $OUT/${cfg}_${E}_$tag.json 2>$OUT/${cfg}_${E}_$tag.err || { echo "$cfg $E $tag FAILED: $(tail -1 $OUT/${cfg}_${E}_$tag.err | cut -c1-200)"; return; }
  python -c "
import json
d=json.load(open('$OUT/${cfg}_${E}_$tag.json')); r=d['roofline']
print('$cfg', $E, '$tag', 'launch_us %.2f'%r['launch_us'], 'frac %.3f'%r['frac'], r['kernel'])
"
}
for rep in a b; do
  for cfg in C4 C4-lean; do
    run $cfg 8192 base_$rep CL_TUNE_NT_STORES=0
    run $cfg 8192 nt1_$rep CL_TUNE_NT_STORES=1
    run $cfg 4096 base_$rep CL_TUNE_NT_STORES=0
    run $cfg 4096 nt1_$rep CL_TUNE_NT_STORES=1
  done
done
for cfg in C4 C4-lean; do
  for bc in 8 16 64; do run $cfg 8192 bc$bc CL_TUNE_B_CHUNK=$bc; done
  run $cfg 8192 fin3 CL_TUNE_FINISH=3
  run $cfg 8192 fin3nt CL_TUNE_FINISH=3 CL_TUNE_NT_STORES=1
done
