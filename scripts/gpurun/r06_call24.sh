#!/bin/bash
# Round 6, twenty-fourth GPU call: the whole GPU suite, smoke() and the default bench line on the tree as it stands.
set -u
OUT=gpurun_out/r06x; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $OUT/gpu_suite.log
python bench.py > $OUT/bench_line.json 2>$OUT/bench_line.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 > $OUT/bench_line_driver_flags.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench_line', 'bench_line_driver_flags'):
    d = json.load(open(f'gpurun_out/r06x/{f}.json')); r = d['roofline']
    print(f, 'value %.4e' % d['value'], 'ms/step %.5f' % d['ms_per_step'], 'frac %.3f' % r['frac'], r['kernel'], 'metric_shape', r['metric_shape']['launch_us'], r['metric_shape']['frac'])
    print('  cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['kind'], 'dropin', d.get('dropin', {}).get('seconds'))
PY
