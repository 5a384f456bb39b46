#!/bin/bash
set -u
OUT=gpurun_out/r03_run10
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_config_sizes.py tests/test_env_gpu.py tests/test_gpu_rollout.py tests/test_gpu_offsets.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 120 python scripts/kpi_cost_probe.py > $OUT/kpi_cost_probe.log 2>&1; cat $OUT/kpi_cost_probe.log
timeout 300 python bench.py --kpi --no-streaming --no-cpu-baseline > $OUT/bench_kpi.json 2>$OUT/bench_kpi.err
timeout 300 python bench.py --config C3 --kpi > $OUT/bench_C3_kpi.json 2>$OUT/bench_C3_kpi.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03_run10/bench_*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], 'frac %.3f' % r['frac'], r['kernel'])
    except Exception as e:
        print(f, 'ERR', e)
PY
