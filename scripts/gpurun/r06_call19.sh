#!/bin/bash
# Round 6, nineteenth GPU call: SQ counters of the HBM-streaming launch (17 x 1 048 576, env-major kernel), both precision models.
set -u
OUT=gpurun_out/r06s; mkdir -p $OUT; export TMPDIR=/tmp
CTR1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
CTR2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_WAVES"
CTR3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACCUM_PREV_HIRES"
for p in chain fp32; do
  n=1
  for C in "$CTR1" "$CTR2" "$CTR3"; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${p}_$n -o run -- python bench.py --envs-per-gpu 1048576 --precision $p --steps 20 --warmup 5 --reps 1 --no-graph --no-cpu-baseline --no-streaming --no-traffic-pass --no-side-entries > /dev/null 2>$OUT/pmc_${p}_$n.log
    n=$((n+1))
  done
  python scripts/pmc_by_kernel.py cl_step $OUT/pmc_${p}_1/*counter_collection.csv $OUT/pmc_${p}_2/*counter_collection.csv $OUT/pmc_${p}_3/*counter_collection.csv > $OUT/sq_streaming_$p.jsonl
  echo "== $p"; cat $OUT/sq_streaming_$p.jsonl
done
