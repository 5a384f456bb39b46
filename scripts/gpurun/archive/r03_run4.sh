#!/bin/bash
set -u
OUT=gpurun_out/r03_run4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
timeout 2400 bash scripts/profile_round.sh r03 > $OUT/profile_round.log 2>&1; echo "profile rc=$?" >> $OUT/profile_round.log
tail -40 $OUT/profile_round.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/prof_r03/bench_*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], r['bound'], 'frac %.3f' % r['frac'], r['kernel'], 'traffic', r.get('traffic'), r.get('traffic_source'))
    except Exception as e:
        print(f, 'ERR', e)
PY
cat gpurun_out/prof_r03/env_step_bench.log
