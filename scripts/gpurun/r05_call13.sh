#!/bin/bash
# Round 5, thirteenth GPU call: the deferred fold with fewer, larger chunks + the new chunk geometry of the thermal kernel -- tests, then timings.
set -u
OUT=gpurun_out/r05n; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_config_sizes.py -m gpu -q --maxfail=10 > $OUT/config_sizes_tests.log 2>&1; echo "rc=$?" >> $OUT/config_sizes_tests.log); tail -12 $OUT/config_sizes_tests.log
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
  for E in 8192 4096 2048 1024; do run C4 $E auto_$rep; done
  run C4 8192 bc32_$rep CL_TUNE_B_CHUNK=32
  run C4 8192 bc64_$rep CL_TUNE_B_CHUNK=64
  run C4 4096 bc64_$rep CL_TUNE_B_CHUNK=64
  run C4 2048 bc64_$rep CL_TUNE_B_CHUNK=64
  run C4-lean 8192 auto_$rep
  run C4-lean 8192 fin1_$rep CL_TUNE_FINISH=1
  run C4-lean 4096 auto_$rep
  run C4-lean 4096 fin1_$rep CL_TUNE_FINISH=1
  run C4-lean 8192 bc64_$rep CL_TUNE_B_CHUNK=64
done
