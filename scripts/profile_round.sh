#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of the same command,
# HBM traffic counters in separate passes, and kernel stats of the other measured shapes.  Outputs under
# gpurun_out/prof_$TAG/; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 2000 --warmup 200 --no-cpu-baseline"
python bench.py > $OUT/bench_line.log 2>$OUT/bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2>$OUT/trace.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- python bench.py --steps 300 --warmup 100 --no-cpu-baseline --no-graph > /dev/null 2>$OUT/pmc_$c.log
done
python scripts/pmc_summary.py $OUT/bench_pmc_summary.json cl_step_ $OUT/pmc_FETCH_SIZE/*counter_collection.csv $OUT/pmc_WRITE_SIZE/*counter_collection.csv > /dev/null
# LSTM stage: instruction mix and matrix-pipe occupancy (SQ block, one pass); observation epilogue: bytes written
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
  --output-format csv -d $OUT/pmc_lstm -o run -- python scripts/lstm_check.py > /dev/null 2>$OUT/pmc_lstm.log
python scripts/pmc_summary.py $OUT/lstm_pmc_summary.json "cl_lstm_kernel<0, true>" $OUT/pmc_lstm/*counter_collection.csv > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_observe -o run -- python scripts/observe_bench.py > /dev/null 2>$OUT/pmc_observe.log
python scripts/pmc_summary.py $OUT/observe_pmc_summary.json cl_observe_kernel $OUT/pmc_observe/*counter_collection.csv > /dev/null
for s in c4_bench rollout_bench lstm_check observe_bench ev_step_bench; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$s -o run -- python scripts/$s.py > $OUT/$s.log 2>$OUT/$s.err
done
ls -R $OUT | head -50
