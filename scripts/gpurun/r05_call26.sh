#!/bin/bash
# Round 5, twenty-sixth GPU call: the load hint on the observation-epilogue instantiation (cl_step_observe_f32), A/B builds; the store hint at 81 920 - 122 880 envs.
set -u
OUT=gpurun_out/r05zz; mkdir -p $OUT; export TMPDIR=/tmp
ALT=$PWD/citylearn_amd/libcl_plainloads.so
for rep in a b; do
  python scripts/step_observe_bench.py 65536 > $OUT/so_default_$rep.log 2>&1; grep "normalised=False" $OUT/so_default_$rep.log | sed 's/^/default: /' | cut -c1-260
  CL_ALT_LIB=$ALT python scripts/step_observe_bench.py 65536 > $OUT/so_plain_$rep.log 2>&1; grep "normalised=False" $OUT/so_plain_$rep.log | sed 's/^/plain:   /' | cut -c1-260
done
run() { # E tag env...
  local E=$1 tag=$2; shift 2
  env "$@" python bench.py --envs-per-gpu $E --no-cpu-baseline --no-streaming --no-traffic-pass --no-chain-entry --steps 1500 --warmup 150 > $OUT/h_${E}_$tag.json 2>$OUT/h_${E}_$tag.err || { echo "$E $tag FAILED"; return; }
  python -c "
import json
d=json.load(open('$OUT/h_${E}_$tag.json')); r=d['roofline']
print($E, '$tag', 'launch_us %.3f'%r['launch_us'], r['kernel'])
"
}
for E in 81920 98304 122880; do run $E nt; run $E plainst CL_TUNE_NT_STORES=2; done
