#!/bin/bash
set -u
rm -rf gpurun_out/prof_r03
timeout 3000 bash scripts/profile_round.sh r03 > gpurun_out/profile_round_r03.log 2>&1; echo "profile rc=$?"
tail -30 gpurun_out/profile_round_r03.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/prof_r03/bench_*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'value %.3e' % d['value'], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], r['bound'], 'frac %.3f' % r['frac'], r['kernel'], 'traffic', r.get('traffic'), r.get('traffic_source'))
        if r.get('hbm_streaming'): print('    streaming', r['hbm_streaming']['launch_us'], r['hbm_streaming']['frac'], r['hbm_streaming']['traffic_source'])
    except Exception as e:
        print(f, 'ERR', e)
PY
cat gpurun_out/prof_r03/env_step_bench.log
