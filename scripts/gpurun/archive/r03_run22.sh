#!/bin/bash
set -u
mkdir -p gpurun_out/r03_run22
timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_bench.py -q -x 2>&1 | tail -5
for k in "" "--kpi"; do
timeout 300 python bench.py --config C3 $k > gpurun_out/r03_run22/bench_C3$k.json 2> gpurun_out/r03_run22/bench_C3$k.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03_run22/bench_*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
PY
