#!/bin/bash
# Round 5, ninth GPU call: the chain's in-step KPI / fused observation launches and the two-envs-per-lane chain rollout -- tests, then timings
# (C5 whole under the chain next to fp32; headline with CLD_KPI under the chain next to fp32), then the GPU suite.
set -u
OUT=gpurun_out/r05i; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_observe.py tests/test_gpu_rollout.py -m gpu -q --maxfail=10 -k "chain" > $OUT/chain_tests.log 2>&1; echo "rc=$?" >> $OUT/chain_tests.log); tail -15 $OUT/chain_tests.log
for f in "" "--f64-chain"; do
  tag=$( [ -z "$f" ] && echo fp32 || echo chain )
  python bench.py --config C5 --no-cpu-baseline $f > $OUT/c5_$tag.json 2>$OUT/c5_$tag.err
  python bench.py --config headline --kpi --no-cpu-baseline --no-traffic-pass --no-streaming --no-chain-entry $f > $OUT/headline_kpi_$tag.json 2>$OUT/headline_kpi_$tag.err
  python -c "
import json
for n in ('c5','headline_kpi'):
    d=json.load(open('$OUT/'+n+'_$tag.json')); r=d['roofline']
    print(n,'$tag','value %.4e'%d['value'],'ms_per_step',d['ms_per_step'],r.get('kernel'),'launch_us',r.get('launch_us'))
"
done
(timeout 1300 python -m pytest tests -m gpu -q --maxfail=20 > $OUT/gpu_suite.log 2>&1; echo "rc=$?" >> $OUT/gpu_suite.log); tail -8 $OUT/gpu_suite.log
