#!/bin/bash
# round 3, GPU call 1: suite + bench plumbing + LSTM variant 32
set -u
OUT=gpurun_out/r03_run1
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2>$OUT/bench_driver_flags.err
CL_BENCH_OVERSUBSCRIBE=1 timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-streaming > $OUT/bench_gpus2_oversub.json 2>$OUT/bench_gpus2_oversub.err
for c in C2 C3 C4 C4-lean C5; do
  timeout 300 python bench.py --config $c > $OUT/bench_$c.json 2>$OUT/bench_$c.err
done
timeout 300 python scripts/lstm_check.py --quick > $OUT/lstm_check.log 2>&1
head -c 600 $OUT/bench_driver_flags.json; echo
cat $OUT/bench_gpus2_oversub.json | head -c 1500; echo
for c in C2 C3 C4 C4-lean C5; do head -c 400 $OUT/bench_$c.json; echo; tail -2 $OUT/bench_$c.err; done
cat $OUT/lstm_check.log
