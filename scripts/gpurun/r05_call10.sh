#!/bin/bash
# Round 5, tenth GPU call: does the power-of-two row stride of BASELINE config 4 (1024 buildings x 8192 envs: 32 KiB rows) cost bandwidth?
# The same kernels at env counts next to the power of two: per-unit time tells (no pitch support in these kernels yet).
set -u
OUT=gpurun_out/r05k; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
for cfg in C4 C4-lean; do
  for E in 8192 8256 8448 1024 1088; do
    python bench.py --config $cfg --envs-per-gpu $E --no-cpu-baseline --steps 600 --warmup 60 > $OUT/${cfg}_${E}_$rep.json 2>$OUT/${cfg}_${E}_$rep.err
    python -c "
import json
d=json.load(open('$OUT/${cfg}_${E}_$rep.json')); r=d['roofline']
print('$cfg', $E, 'rep$rep', 'launch_us %.2f'%r['launch_us'], 'ns/kunit %.3f'%(r['launch_us']*1e6/(1024*$E)), 'frac %.3f'%r['frac'], r['kernel'])
"
  done
done
done
