#!/bin/bash
# vector-ALU counters of the thermal step kernels: is T9 / T9 + KPIs / C4 / the headline bound by instruction issue?
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03_run25
mkdir -p $OUT
CTR="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
run() { name=$1; shift; rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc_$name -o run -- "$@" > /dev/null 2>$OUT/pmc_$name.log; python scripts/pmc_by_kernel.py cl_step $OUT/pmc_$name/*counter_collection.csv > $OUT/${name}_sq_by_kernel.jsonl; cat $OUT/${name}_sq_by_kernel.jsonl; }
run t9 python bench.py --config T9 --steps 60 --warmup 20 --reps 1 --no-graph --no-cpu-baseline
run kpi_t9 python bench.py --config T9 --kpi --steps 60 --warmup 20 --reps 1 --no-graph --no-cpu-baseline
run c4 python bench.py --config C4 --steps 60 --warmup 20 --reps 1 --no-graph
run headline python bench.py --steps 60 --warmup 20 --reps 1 --no-graph --no-cpu-baseline --no-streaming
