#!/bin/bash
# Round 5, eighteenth GPU call: cl_step_lean_chunk_kernel -- equivalence tests, then A/B against cl_step_kernel (lean_variant = 16) on the battery + PV
# config-4 shapes, alternating, and the chunk sizes once more (more buildings per wave may pay now).
set -u
OUT=gpurun_out/r05s; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_config_sizes.py -m gpu -q --maxfail=10 > $OUT/config_sizes_tests.log 2>&1; echo "rc=$?" >> $OUT/config_sizes_tests.log); tail -12 $OUT/config_sizes_tests.log
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
  for E in 1024 2048 4096 8192; do
    run C4-lean $E new_$rep
    run C4-lean $E old_$rep CL_TUNE_LEAN_VARIANT=16
  done
done
for bc in 16 64 128; do run C4-lean 8192 bc$bc CL_TUNE_B_CHUNK=$bc; done
for bc in 16 64; do run C4-lean 1024 bc$bc CL_TUNE_B_CHUNK=$bc; done
run C4-lean 8192 nt1 CL_TUNE_NT_STORES=1
run C4-lean 8192 bc64nt1 CL_TUNE_B_CHUNK=64 CL_TUNE_NT_STORES=1
