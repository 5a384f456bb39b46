#!/bin/bash
set -u
OUT=gpurun_out/r03_run9
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
timeout 120 python scripts/kpi_cost_probe.py > $OUT/kpi_cost_probe.log 2>&1; cat $OUT/kpi_cost_probe.log
timeout 300 python bench.py --config C3 > $OUT/bench_C3.json 2>$OUT/bench_C3.err
timeout 300 python bench.py --config C3 --kpi > $OUT/bench_C3_kpi.json 2>$OUT/bench_C3_kpi.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03_run9/bench_*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
    except Exception as e:
        print(f, 'ERR', e)
PY
