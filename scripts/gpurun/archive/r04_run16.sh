#!/bin/bash
# final tree: smoke(), the driver's bench command, the default bench command
set -u
O=gpurun_out/r04_run16; mkdir -p $O
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "rc $?"
python -c "
import json; d=json.load(open('$O/bench_driver_flags.json')); r=d['roofline']
print('value %.4e ms/step %.5f launch_us %.2f frac %.3f residency %s' % (d['value'], d['ms_per_step'], r['launch_us'], r['frac'], r.get('residency')))
print('streaming', r['hbm_streaming']['launch_us'], r['hbm_streaming']['frac'], 'traffic', r['traffic'], r['traffic_source'])
c=d['cpu_baseline']; print('ref', c['reference']['value'], c['reference']['measured'], '| port', c['value'])"
