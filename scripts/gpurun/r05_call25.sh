#!/bin/bash
# Round 5, twenty-fifth GPU call: the final tree -- GPU suite, smoke, the default bench line and the driver's flags, mid-size batches under the new selection rule.
set -u
OUT=gpurun_out/r05_final; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1300 python -m pytest tests -m gpu -q --maxfail=20 > $OUT/gpu_suite.log 2>&1; echo "rc=$?" >> $OUT/gpu_suite.log); tail -5 $OUT/gpu_suite.log
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py > $OUT/bench_line.json 2>$OUT/bench_line.err
python -c "import json;d=json.load(open('$OUT/bench_line.json'));r=d['roofline'];print('value %.4e'%d['value'],'ms_per_step',d['ms_per_step'],'launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],'traffic',r['traffic'],'stream',r['hbm_streaming']['launch_us'],r['hbm_streaming']['frac'],'chain',r['f64_chain']['launch_us'],'cpu',d['cpu_baseline']['kind'],d['cpu_baseline']['value'])"
python bench.py --steps 20 --warmup 5 > $OUT/bench_line_driver_flags.json 2>/dev/null
python -c "import json;d=json.load(open('$OUT/bench_line_driver_flags.json'));print('driver flags value %.4e'%d['value'],d['ms_per_step'])"
for E in 98304 114688; do python bench.py --envs-per-gpu $E --no-cpu-baseline --no-streaming --no-traffic-pass --no-chain-entry > $OUT/h_$E.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/h_$E.json'));r=d['roofline'];print($E,'launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],r['kernel'])"; done
