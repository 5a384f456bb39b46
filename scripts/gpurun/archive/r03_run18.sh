#!/bin/bash
# fused thermal KPI: whole GPU suite + T9 bench lines
set -u
mkdir -p gpurun_out/r03_run18
true
for k in "" "--kpi"; do
  timeout 300 python bench.py --config T9 $k --no-cpu-baseline > gpurun_out/r03_run18/bench_T9$k.json 2> gpurun_out/r03_run18/bench_T9$k.err; echo "T9 $k rc=$?"
done
timeout 300 python bench.py --kpi --no-streaming --no-cpu-baseline > gpurun_out/r03_run18/bench_kpi.json 2> gpurun_out/r03_run18/bench_kpi.err
timeout 300 python bench.py --kpi --config C3 > gpurun_out/r03_run18/bench_kpi_C3.json 2> gpurun_out/r03_run18/bench_kpi_C3.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03_run18/bench_*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'value %.3e' % d['value'], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], r['bound'], 'frac %.3f' % r['frac'], r['kernel'], r.get('algorithmic_bytes_per_unit'))
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 300 python scripts/kpi_cost_probe.py > gpurun_out/r03_run18/kpi_in_step_probe.log 2>&1
