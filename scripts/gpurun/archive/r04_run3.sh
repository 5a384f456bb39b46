#!/bin/bash
# rocprof durations of the C4 shard kernels, deferred vs launch-per-step; observation epilogue with the one-round-trip row-wise kernel
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_run3; mkdir -p $O
for c in C4-lean C4; do
  for f in 3 1; do
    CL_TUNE_FINISH=$f rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${c}_$f -o run -- python bench.py --config $c --reps 1 > /dev/null 2>$O/trace_${c}_$f.log
    echo "== $c finish=$f"; cut -c1-140 $O/trace_${c}_$f/*kernel_stats.csv | head -3
  done
done
timeout 900 python -m pytest tests/test_gpu_observe.py -x -q 2>&1 | tail -3
timeout 600 python scripts/observe_bench.py > $O/observe_bench.log 2>$O/observe_bench.err
cat $O/observe_bench.log
