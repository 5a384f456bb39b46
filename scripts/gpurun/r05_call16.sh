#!/bin/bash
# Round 5, sixteenth GPU call: the K = 20 timed region's host overhead (wall 8.65 vs kernel 7.55 us per step).
set -u
OUT=gpurun_out/r05q; mkdir -p $OUT; export TMPDIR=/tmp
python scripts/r05_k20_overhead.py 20 2>&1 | tee $OUT/k20_default.log | tail -4
HSA_ENABLE_INTERRUPT=0 python scripts/r05_k20_overhead.py 20 2>&1 | tee $OUT/k20_poll.log | tail -4
python scripts/r05_k20_overhead.py 100 2>&1 | tee $OUT/k100_default.log | tail -2
HSA_ENABLE_INTERRUPT=0 python scripts/r05_k20_overhead.py 100 2>&1 | tee $OUT/k100_poll.log | tail -2
