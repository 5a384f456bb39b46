#!/bin/bash
# Round 6, third GPU call: the tests that failed in call 2 (fixed), the bench tests under the new line structure, the default bench line as the
# driver runs it, and the kernel-selection map (scripts/r06_cliffs.py).
set -u
OUT=gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_check.py tests/test_gpu_config_sizes.py tests/test_gpu_parity.py -m gpu -q -k "check or distinct_actions or chunked_grid or lean_chunk or c4_shard_under or kernel_selection" > $OUT/fixed_tests.log 2>&1
echo "fixed tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/fixed_tests.log | tail -20
timeout 1500 python -m pytest tests/test_gpu_bench.py -m gpu -q > $OUT/bench_tests.log 2>&1
echo "bench tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/bench_tests.log | tail -20
(time timeout 900 python bench.py) > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06c/bench_default.json'))
r = d['roofline']
print('value %.4g  ms_per_step %.6f  dtype %s' % (d['value'], d['ms_per_step'], d['dtype']))
print('roofline: frac %.3f  %s  %.2f us  traffic %s' % (r['frac'], r['kernel'], r['launch_us'], r.get('traffic')))
m = r.get('metric_shape', {})
print('metric_shape: frac %.3f %s %.3f us traffic %s' % (m.get('frac', 0), m.get('kernel'), m.get('launch_us', 0), m.get('traffic')))
print('fp32:', json.dumps(r.get('fp32_map'))[:600])
print('dropin:', d.get('dropin'))
cb = d.get('cpu_baseline', {})
print('cpu:', cb.get('value'), cb.get('cores'), cb.get('sample'), cb.get('measured'), (cb.get('c1_single_process') or {}).get('seconds'))
PY
timeout 1700 python scripts/r06_cliffs.py $OUT/cliffs_chain.jsonl chain > $OUT/cliffs_chain.log 2>&1; echo "cliffs rc=$?"; tail -5 $OUT/cliffs_chain.log
