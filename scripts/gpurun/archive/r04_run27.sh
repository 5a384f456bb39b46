#!/bin/bash
mkdir -p gpurun_out/r04_run27
python bench.py > gpurun_out/r04_run27/bench_line.json 2>/dev/null; echo rc $?
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_run27/bench_line_driver_flags.json 2>/dev/null; echo rc $?
python -c "
import json
for f in ('bench_line','bench_line_driver_flags'):
    d=json.load(open('gpurun_out/r04_run27/'+f+'.json')); r=d['roofline']; print(f, 'value %.3e ms/step %.5f launch %.2f frac %.3f stream %.3f traffic %.2f MB (%s)' % (d['value'], d['ms_per_step'], r['launch_us'], r['frac'], r['hbm_streaming']['frac'], r['traffic']/1e6, r['traffic_source'][:20]), 'ref', round(d['cpu_baseline']['reference']['value']))"
