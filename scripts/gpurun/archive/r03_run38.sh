#!/bin/bash
# lone-rank RCCL runs in a row after moving the bracket's barrier off the capture stream
set -u
mkdir -p gpurun_out/r03_run38
fails=0
for i in $(seq 1 36); do
  CL_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streaming > /tmp/out.txt 2> /tmp/err.txt; rc=$?
  if [ "$(wc -l < /tmp/out.txt)" != "1" ] || [ $rc != 0 ]; then fails=$((fails+1)); cp /tmp/err.txt gpurun_out/r03_run38/err_$i.txt; fi
done
echo "barrier on the default stream: $fails of 36 lone-rank RCCL runs failed" | tee gpurun_out/r03_run38/summary.log
python -c "
import json; d=json.loads(open('/tmp/out.txt').read()); print(d['control_backend'], d['ms_per_step'], d['roofline']['launch_us'])"
