#!/bin/bash
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_run22_line.json 2>gpurun_out/r04_run22.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('gpurun_out/r04_run22_line.json')); r=d['roofline']; h=r['hbm_streaming']
print(r['traffic'], r['traffic_source'][:40]); print(h['traffic'], h['traffic_source'][:60], h.get('traffic_live_error'), h['launch_us'], h['frac'])
print(h['traffic']/(h['units_per_launch']*36.94117647))"
tail -3 gpurun_out/r04_run22.err
