#!/bin/bash
set -u
OUT=gpurun_out/r03_run6
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
timeout 600 python scripts/lstm_generic_bench.py > $OUT/lstm_generic_bench.log 2>&1; cat $OUT/lstm_generic_bench.log
timeout 300 python scripts/finish_stress.py > $OUT/finish_stress.log 2>&1; tail -4 $OUT/finish_stress.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2>$OUT/bench_driver_flags.err; head -c 400 $OUT/bench_driver_flags.json; echo
