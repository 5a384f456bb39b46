#!/bin/bash
set -u
O=gpurun_out/r04_run20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -4
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('$O/bench_driver_flags.json')); r=d['roofline']
print(r['traffic'], r['traffic_source'], r.get('traffic_live_error'), r.get('traffic_committed_file'))"
