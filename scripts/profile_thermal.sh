#!/bin/bash
# Thermal / outage step kernel (cl_step_kernel<.., FULL>) evidence on the GPU box: un-profiled timings, rocprofv3 kernel stats,
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and the SQ set, at the three shapes VERDICT r01 names:
# 2023 schema 3 x 65 536 (C3), 2020 schema 9 x 65 536, C4 shard 1024 buildings x 1024 envs.  Outputs under gpurun_out/thermal_$TAG/.
set -u
TAG=${1:-r02}
OUT=gpurun_out/thermal_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python scripts/c4_bench.py > $OUT/c4_bench_unprofiled.log 2>$OUT/c4_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python scripts/c4_bench.py > $OUT/c4_bench_under_rocprof.log 2>$OUT/trace.log
for cfg in "g2023_p2 65536 0" "g2020_cz1 65536 0" "g2020_cz1 1024 1024"; do
  set -- $cfg
  name=$1_$2_$3
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${name}_$c -o run -- python scripts/run_cfg.py $1 $2 $3 60 > /dev/null 2>$OUT/pmc_${name}_$c.log
  done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
    --output-format csv -d $OUT/pmc_${name}_SQ -o run -- python scripts/run_cfg.py $1 $2 $3 60 > /dev/null 2>$OUT/pmc_${name}_SQ.log
  python scripts/pmc_summary.py $OUT/${name}_pmc_summary.json "cl_step_kernel" $OUT/pmc_${name}_FETCH_SIZE/*counter_collection.csv \
    $OUT/pmc_${name}_WRITE_SIZE/*counter_collection.csv $OUT/pmc_${name}_SQ/*counter_collection.csv > /dev/null
done
cp $OUT/trace/*kernel_stats.csv $OUT/c4_bench_kernel_stats.csv 2>/dev/null
ls $OUT
