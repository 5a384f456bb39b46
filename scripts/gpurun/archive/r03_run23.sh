#!/bin/bash
set -u
mkdir -p gpurun_out/r03_run23
timeout 900 python -m pytest tests/test_gpu_config_sizes.py tests/test_env_gpu.py tests/test_gpu_offsets.py tests/test_gpu_rollout.py -q -x 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --kpi --no-streaming --no-cpu-baseline > gpurun_out/r03_run23/bench_kpi_$i.json 2> gpurun_out/r03_run23/bench_kpi.err; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03_run23/bench_*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], 'frac %.3f' % r['frac'], r['kernel'])
PY
