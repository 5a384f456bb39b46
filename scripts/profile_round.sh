#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun): bench lines (default and the driver's --steps 20 --warmup 5), rocprofv3
# kernel stats of the same command, HBM traffic counters in separate passes, thermal-kernel evidence, mode-B SQ counters,
# observation / EV step timings.  Outputs under gpurun_out/prof_$TAG/; copy what should be judged into profiles/.
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-streaming"
python bench.py > $OUT/bench_line.json 2>$OUT/bench_line.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_line_driver_flags.json 2>$OUT/bench_line_driver_flags.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2>$OUT/trace.log
cp $OUT/trace/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- python bench.py --steps 300 --warmup 100 --no-cpu-baseline --no-graph --no-streaming > /dev/null 2>$OUT/pmc_$c.log
done
python scripts/pmc_summary.py $OUT/bench_pmc_summary.json cl_step_ $OUT/pmc_FETCH_SIZE/*counter_collection.csv $OUT/pmc_WRITE_SIZE/*counter_collection.csv > /dev/null
# thermal / outage kernel: un-profiled sweep lines, kernel stats, traffic + SQ counters per shape
python scripts/c4_bench.py > $OUT/c4_bench_unprofiled.log 2>$OUT/c4_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_trace -o run -- python scripts/c4_bench.py > $OUT/c4_bench_under_rocprof.log 2>$OUT/c4_trace.log
cp $OUT/c4_trace/*kernel_stats.csv $OUT/c4_bench_kernel_stats.csv 2>/dev/null
for cfg in "g2023_p2 65536 0" "g2020_cz1 65536 0" "g2020_cz1 1024 1024"; do
  set -- $cfg
  name=$1_$2_$3
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${name}_$c -o run -- python scripts/run_cfg.py $1 $2 $3 60 > /dev/null 2>$OUT/pmc_${name}_$c.log
  done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
    --output-format csv -d $OUT/pmc_${name}_SQ -o run -- python scripts/run_cfg.py $1 $2 $3 60 > /dev/null 2>$OUT/pmc_${name}_SQ.log
  python scripts/pmc_summary.py $OUT/thermal_${name}_pmc_summary.json "cl_step_full" $OUT/pmc_${name}_FETCH_SIZE/*counter_collection.csv \
    $OUT/pmc_${name}_WRITE_SIZE/*counter_collection.csv $OUT/pmc_${name}_SQ/*counter_collection.csv > /dev/null
done
# mode B (fused rollout): VALU instruction counts and vector-ALU busy time
python scripts/rollout_bench.py > $OUT/rollout_bench_unprofiled.log 2>$OUT/rollout_bench.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/pmc_rollout -o run -- python scripts/rollout_bench.py > /dev/null 2>$OUT/pmc_rollout.log
python scripts/pmc_by_kernel.py cl_rollout_kernel $OUT/pmc_rollout/*counter_collection.csv > $OUT/rollout_pmc_by_kernel.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rollout_trace -o run -- python scripts/rollout_bench.py > /dev/null 2>$OUT/rollout_trace.log
cp $OUT/rollout_trace/*kernel_stats.csv $OUT/rollout_kernel_stats.csv 2>/dev/null
for s in observe_bench ev_step_bench lstm_check; do
  python scripts/$s.py > $OUT/${s}_unprofiled.log 2>$OUT/$s.err
done
ls $OUT | head -60
