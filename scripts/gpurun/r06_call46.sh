#!/bin/bash
# Round 6, forty-sixth GPU call: the env-major / building-major rule after the alternating sweep -- selection tests, the observe and offset tests, the default bench line.
set -u
OUT=gpurun_out/r06z8; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_config_sizes.py tests/test_gpu_observe.py tests/test_gpu_offsets.py tests/test_gpu_bench.py -m gpu -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -20
python bench.py > $OUT/bench_line.json 2>$OUT/bench_line.err
python bench.py --precision fp32 --no-cpu-baseline --no-traffic-pass > $OUT/bench_fp32.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench_line', 'bench_fp32'):
    d = json.load(open(f'gpurun_out/r06z8/{f}.json')); r = d['roofline']
    print(f, 'value %.4e' % d['value'], 'frac %.3f' % r['frac'], 'launch_us %.2f' % r['launch_us'], r['kernel'], 'traffic', r.get('traffic'), '| metric', r['metric_shape']['launch_us'], r['metric_shape']['frac'])
    if 'fp32_map' in r: print('   fp32_map', {k: (v.get('launch_us'), v.get('frac', v.get('frac_of_hbm_peak'))) for k, v in r['fp32_map'].items() if isinstance(v, dict)})
PY
