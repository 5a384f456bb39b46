#!/bin/bash
# where does the deferred fold's time go?  finish = 3 full fold; 13 no global load; 14 load + stash, no add-up; 15 double buffer + marker only; 1 second launch
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_run5; mkdir -p $O
for rep in 1 2; do
for c in C4-lean C4; do
  for f in 3 13 14 15 1; do
    CL_TUNE_FINISH=$f rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${c}_${f}_$rep -o run -- python bench.py --config $c --reps 1 --steps 2000 --warmup 100 > /dev/null 2>$O/trace.log
    echo "$c finish=$f rep $rep: $(grep -m1 'cl_step' $O/trace_${c}_${f}_$rep/*kernel_stats.csv | awk -F, '{print $(NF-4)}')"
  done
done
done
