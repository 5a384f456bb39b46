#!/bin/bash
set -u
O=gpurun_out/r04_run8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "f64 or north_star" 2>&1 | tail -3
timeout 600 python scripts/f64_cost.py > $O/f64_cost.log 2>&1; cat $O/f64_cost.log
timeout 300 python bench.py --f64-maps --no-streaming --no-cpu-baseline > $O/bench_f64_maps.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_f64_maps.json')); r=d['roofline']; print('f64 headline', r['launch_us'], r['frac'], r['kernel'])"
