#!/bin/bash
# Round 5, eighth GPU call: the GPU suite and the default bench line on the round's final tree.
set -u
OUT=gpurun_out/r05h; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 > $OUT/gpu_suite_final.log 2>&1; echo "rc=$?" >> $OUT/gpu_suite_final.log); tail -25 $OUT/gpu_suite_final.log
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py > $OUT/bench_line_final.json 2>$OUT/bench_line_final.err
python -c "import json;d=json.load(open('$OUT/bench_line_final.json'));r=d['roofline'];print('value %.4e'%d['value'],'ms_per_step',d['ms_per_step'],'launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],'traffic',r['traffic'],'stream',r['hbm_streaming']['launch_us'],r['hbm_streaming']['frac'],'chain',r['f64_chain']['launch_us'],'cpu',d['cpu_baseline']['kind'],d['cpu_baseline']['value'])"
python bench.py --steps 20 --warmup 5 > $OUT/bench_line_final_driver_flags.json 2>/dev/null
python -c "import json;d=json.load(open('$OUT/bench_line_final_driver_flags.json'));print('driver flags value %.4e'%d['value'],d['ms_per_step'])"
