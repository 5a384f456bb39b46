#!/bin/bash
set -u
OUT=gpurun_out/r03_run11
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 120 python scripts/kpi_cost_probe.py > $OUT/kpi_cost_probe.log 2>&1; cat $OUT/kpi_cost_probe.log
python bench.py --kpi --no-streaming --no-cpu-baseline > $OUT/bench_kpi.json 2>$OUT/bench_kpi.err
python bench.py --config C3 --kpi > $OUT/bench_kpi_C3.json 2>$OUT/bench_kpi_C3.err
python bench.py --config C3 > $OUT/bench_C3.json 2>$OUT/bench_C3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c3 -o run -- python bench.py --config C3 --reps 1 > /dev/null 2>$OUT/trace_c3.log
cp $OUT/trace_c3/*kernel_stats.csv $OUT/c3_kernel_stats.csv
python scripts/check_profiles.py $OUT/bench_C3.json $OUT/c3_kernel_stats.csv
python bench.py --steps 20 --warmup 5 > $OUT/bench_line_driver_flags.json 2>$OUT/bench_line_driver_flags.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03_run11/bench_*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], 'frac %.3f' % r['frac'], r['kernel'])
    except Exception as e:
        print(f, 'ERR', e)
PY
head -4 $OUT/c3_kernel_stats.csv
