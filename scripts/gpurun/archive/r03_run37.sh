#!/bin/bash
# lone-rank RCCL runs in a row: the watchdog-vs-capture abort (global capture mode) and the fix (thread-local)
set -u
mkdir -p gpurun_out/r03_run37
for mode in thread_local global; do
  fails=0
  for i in $(seq 1 14); do
    CL_BENCH_CAPTURE_MODE=$mode CL_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streaming > /tmp/out.txt 2> /tmp/err.txt; rc=$?
    if [ "$(wc -l < /tmp/out.txt)" != "1" ] || [ $rc != 0 ]; then fails=$((fails+1)); cp /tmp/err.txt gpurun_out/r03_run37/err_${mode}_$i.txt; fi
  done
  echo "capture mode $mode: $fails of 14 runs failed"
done | tee gpurun_out/r03_run37/summary.log
grep -h -m3 -i "capture\|error" gpurun_out/r03_run37/err_*.txt 2>/dev/null | cut -c1-300 | head -8
