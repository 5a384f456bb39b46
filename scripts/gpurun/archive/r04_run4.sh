#!/bin/bash
# deferred finish with the fold load hoisted to the kernel top; persistent observation kernel
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_run4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config_sizes.py tests/test_gpu_observe.py -x -q -k "c4 or observ or obs" 2>&1 | tail -4
for c in C4-lean C4; do
  for f in 3 1; do
    CL_TUNE_FINISH=$f rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${c}_$f -o run -- python bench.py --config $c --reps 1 > $O/bench_${c}_${f}_under_rocprof.json 2>$O/trace_${c}_$f.log
    echo "== $c finish=$f"; cut -c1-140 $O/trace_${c}_$f/*kernel_stats.csv | head -3
    CL_TUNE_FINISH=$f timeout 300 python bench.py --config $c > $O/bench_${c}_finish$f.json 2> $O/bench_${c}_finish$f.err
    python -c "
import json; d=json.load(open('$O/bench_${c}_finish$f.json')); r=d['roofline']
print('$c finish=$f', 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], 'frac %.3f' % r['frac'], r['kernel'])"
  done
done
timeout 600 python scripts/observe_bench.py > $O/observe_bench.log 2>$O/observe_bench.err
grep -v "compact form\|all-exogenous" $O/observe_bench.log; tail -3 $O/observe_bench.err
