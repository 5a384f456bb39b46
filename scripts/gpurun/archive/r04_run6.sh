#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_run6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config_sizes.py tests/test_gpu_observe.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for c in C4-lean C4; do
  for f in 3 1; do
    CL_TUNE_FINISH=$f rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${c}_${f}_$rep -o run -- python bench.py --config $c --reps 1 --steps 2000 --warmup 100 > $O/line_${c}_${f}_$rep.json 2>$O/trace.log
    echo "$c finish=$f rep $rep: kernel $(grep -m1 'cl_step' $O/trace_${c}_${f}_$rep/*kernel_stats.csv | awk -F, '{print $(NF-4)}') ns; line $(python -c "import json; d=json.load(open('$O/line_${c}_${f}_$rep.json')); print('%.2f us frac %.3f' % (d['roofline']['launch_us'], d['roofline']['frac']))")"
  done
done
done
