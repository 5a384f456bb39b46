#!/bin/bash
# round 3, GPU call 2: f64 maps (parity + cost), in-kernel finish of chunked districts (stress + A/B), C3 with the 7-transcendental cell
set -u
OUT=gpurun_out/r03_run2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/finish_stress.py > $OUT/finish_stress.log 2>&1; echo "rc=$?" >> $OUT/finish_stress.log
tail -5 $OUT/finish_stress.log
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
for c in C4 C4-lean; do
  timeout 300 python bench.py --config $c --steps 2000 > $OUT/bench_$c.json 2>$OUT/bench_$c.err
  CL_TUNE_FINISH=1 timeout 300 python bench.py --config $c --steps 2000 > $OUT/bench_${c}_two_launch.json 2>$OUT/bench_${c}_two_launch.err
done
timeout 300 python bench.py --config C3 > $OUT/bench_C3.json 2>$OUT/bench_C3.err
timeout 600 python scripts/f64_cost.py > $OUT/f64_cost.log 2>&1
timeout 300 python bench.py --f64-maps --steps 2000 --no-cpu-baseline > $OUT/bench_headline_f64.json 2>$OUT/bench_headline_f64.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03_run2/bench_*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], 'frac %.3f' % r['frac'], r['kernel'])
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $OUT/f64_cost.log
