#!/bin/bash
# deferred finish of the C4 shard's district sums: tests + A/B against the launch per step
set -u
O=gpurun_out/r04_run2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config_sizes.py -x -q -k "c4" > $O/c4_tests.log 2>&1; tail -15 $O/c4_tests.log
for c in C4-lean C4; do
  for f in 3 1; do
    CL_TUNE_FINISH=$f timeout 300 python bench.py --config $c > $O/bench_${c}_finish$f.json 2> $O/bench_${c}_finish$f.err
    python -c "
import json; d=json.load(open('$O/bench_${c}_finish$f.json')); r=d['roofline']
print('$c finish=$f', 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], 'frac %.3f' % r['frac'], r['kernel'])"
  done
done
CL_TUNE_FINISH=3 timeout 300 python bench.py --config C4-lean --steps 20 --warmup 5 > $O/bench_C4-lean_driver.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_C4-lean_driver.json')); print('driver flags C4-lean', d['ms_per_step'], d['roofline']['launch_us'])"
