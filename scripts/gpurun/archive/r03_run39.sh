#!/bin/bash
# lone-rank RCCL, the default bench INCLUDING the streaming measurement (captures after the first measurement's all-reduces)
set -u
mkdir -p gpurun_out/r03_run39
fails=0
for i in $(seq 1 12); do
  CL_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > /tmp/out.txt 2> /tmp/err.txt; rc=$?
  if [ "$(wc -l < /tmp/out.txt)" != "1" ] || [ $rc != 0 ]; then fails=$((fails+1)); cp /tmp/err.txt gpurun_out/r03_run39/err_$i.txt; fi
done
echo "default bench with the streaming entry, RCCL up: $fails of 12 runs failed" | tee gpurun_out/r03_run39/summary.log
python -c "
import json; d=json.loads(open('/tmp/out.txt').read()); print(d['control_backend'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['hbm_streaming']['frac'])"
timeout 600 python -m pytest tests/test_gpu_bench.py -q 2>&1 | tail -2
