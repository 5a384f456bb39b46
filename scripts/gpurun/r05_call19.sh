#!/bin/bash
# Round 5, nineteenth GPU call: cl_step_lean_chunk_kernel with plain prefetch loads -- A/B against cl_step_kernel (lean_variant = 16), alternating.
set -u
OUT=gpurun_out/r05t; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_config_sizes.py -m gpu -q --maxfail=10 > $OUT/config_sizes_tests.log 2>&1; echo "rc=$?" >> $OUT/config_sizes_tests.log); tail -4 $OUT/config_sizes_tests.log
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
  for E in 512 1024 2048 3072 4096 8192; do
    run C4-lean $E new_$rep
    run C4-lean $E old_$rep CL_TUNE_LEAN_VARIANT=16
  done
done
run C4-lean 2048 bc16_new CL_TUNE_B_CHUNK=16
run C4-lean 2048 bc16_old CL_TUNE_B_CHUNK=16 CL_TUNE_LEAN_VARIANT=16
