#!/bin/bash
# round 4, first contact: GPU suite on the tree as it is + the default bench line with the reference timed live on the box
set -u
O=gpurun_out/r04_run1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?" >> $O/gpu_suite.log
tail -3 $O/gpu_suite.log
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('$O/bench_line.json'))
print('value', d['value'], 'frac', d['roofline']['frac'], 'stream', d['roofline']['hbm_streaming']['frac'])
r=d['cpu_baseline']['reference']; print({k: r.get(k) for k in ('value','cores','host','host_gpu','measured','sample','seconds_including_env_construction','live_error')})
print('port', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
