#!/bin/bash
# Round 6, fourth GPU call: the whole GPU suite after the selection-rule changes; the HBM-streaming shape (17 x 1 048 576) A/B -- plain env-major kernel,
# narrower workgroups, the persistent pipelined kernel at several grids -- for both precision models; the selection map again.
set -u
OUT=gpurun_out/r06d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench.py > $OUT/suite.log 2>&1
echo "suite rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/suite.log | tail -30
run() { # tag precision env...
  local tag=$1 prec=$2; shift 2
  env "$@" python bench.py --envs-per-gpu 1048576 --precision $prec --no-cpu-baseline --no-streaming --no-traffic-pass --no-side-entries --steps 30 --warmup 5 --reps 3 > $OUT/s_${prec}_$tag.json 2>$OUT/s_${prec}_$tag.err || { echo "$prec $tag FAILED: $(tail -1 $OUT/s_${prec}_$tag.err | cut -c1-200)"; return; }
  python -c "
import json
d=json.load(open('$OUT/s_${prec}_$tag.json')); r=d['roofline']
print('$prec', '$tag', 'launch_us %.2f'%r['launch_us'], 'frac %.3f'%r['frac'], r['kernel'])
"
}
for rep in 1 2; do
for prec in chain fp32; do
  run plain$rep $prec CL_TUNE_ENVMAJOR=1
  run pipe768_$rep $prec CL_TUNE_ENVMAJOR=3
  run pipe512_$rep $prec CL_TUNE_ENVMAJOR=3 CL_TUNE_B_CHUNK=512
  run pipe1024_$rep $prec CL_TUNE_ENVMAJOR=3 CL_TUNE_B_CHUNK=1024
done
run waves2_$rep chain CL_TUNE_ENVMAJOR=1 CL_TUNE_NW=2
run waves1_$rep chain CL_TUNE_ENVMAJOR=1 CL_TUNE_NW=1
done
timeout 1500 python scripts/r06_cliffs.py $OUT/cliffs_chain.jsonl chain > $OUT/cliffs_chain.log 2>&1; echo "cliffs rc=$?"; grep -c . $OUT/cliffs_chain.jsonl
