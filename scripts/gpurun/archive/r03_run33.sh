#!/bin/bash
set -u
mkdir -p gpurun_out/r03_run33
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r03_run33/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py > gpurun_out/r03_run33/bench_default.json 2> gpurun_out/r03_run33/bench_default.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('gpurun_out/r03_run33/bench_default.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['launch_us'], r['frac'], r['kernel'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
