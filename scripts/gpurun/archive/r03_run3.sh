#!/bin/bash
set -u
OUT=gpurun_out/r03_run3
mkdir -p $OUT
timeout 300 python scripts/finish_stress.py > $OUT/finish_stress.log 2>&1; echo "rc=$?" >> $OUT/finish_stress.log
cat $OUT/finish_stress.log
for c in C4 C4-lean; do
  timeout 300 python bench.py --config $c --steps 2000 > $OUT/bench_$c.json 2>$OUT/bench_$c.err
  CL_TUNE_FINISH=1 timeout 300 python bench.py --config $c --steps 2000 > $OUT/bench_${c}_two_launch.json 2>$OUT/bench_${c}_two_launch.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03_run3/bench_*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], 'frac %.3f' % r['frac'], r['kernel'])
    except Exception as e:
        print(f, 'ERR', e)
PY
