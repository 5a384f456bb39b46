#!/bin/bash
# Round 5, twenty-second GPU call: the tree with the load hint chosen by instantiation -- GPU suite, default bench line, the lean variants, user-level timings.
set -u
OUT=gpurun_out/r05w; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1300 python -m pytest tests -m gpu -q --maxfail=20 > $OUT/gpu_suite.log 2>&1; echo "rc=$?" >> $OUT/gpu_suite.log); tail -5 $OUT/gpu_suite.log
python bench.py > $OUT/bench_line.json 2>$OUT/bench_line.err
python -c "import json;d=json.load(open('$OUT/bench_line.json'));r=d['roofline'];print('value %.4e'%d['value'],'ms_per_step',d['ms_per_step'],'launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],'traffic',r['traffic'],'stream',r['hbm_streaming']['launch_us'],r['hbm_streaming']['frac'],'chain',r['f64_chain']['launch_us'],'cpu',d['cpu_baseline']['kind'],d['cpu_baseline']['value'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_driver_flags.json 2>/dev/null
python -c "import json;d=json.load(open('$OUT/bench_line_driver_flags.json'));print('driver flags value %.4e'%d['value'],d['ms_per_step'])"
for c in C2 C4-lean T9; do python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/bench_$c.json'));r=d['roofline'];print('$c','launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],r['kernel'])"; done
python bench.py --kpi --no-streaming --no-cpu-baseline --no-traffic-pass > $OUT/bench_kpi.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/bench_kpi.json'));r=d['roofline'];print('kpi','launch_us %.2f'%r['launch_us'],r['kernel'])"
for s in env_step_bench observe_bench ev_step_bench; do timeout 400 python scripts/$s.py > $OUT/${s}.log 2>$OUT/$s.err; done
grep -h "step+observe\|compact form" $OUT/observe_bench.log | cut -c1-220 | head -12
tail -4 $OUT/ev_step_bench.log | cut -c1-250
grep -h "2022" $OUT/env_step_bench.log | cut -c1-250 | head -8
