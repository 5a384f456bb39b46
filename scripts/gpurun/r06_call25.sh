#!/bin/bash
set -u
OUT=gpurun_out/r06y; mkdir -p $OUT; export TMPDIR=/tmp
CTR1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
CTR2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_WAVES"
CTR3="SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_BUSY_CU_CYCLES"
n=1
for C in "$CTR1" "$CTR2" "$CTR3"; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_c3_$n -o run -- python bench.py --config C3 --steps 40 --warmup 10 --reps 1 --no-graph > /dev/null 2>$OUT/pmc_c3_$n.log
  n=$((n+1))
done
python scripts/pmc_by_kernel.py cl_lstm $OUT/pmc_c3_1/*counter_collection.csv $OUT/pmc_c3_2/*counter_collection.csv $OUT/pmc_c3_3/*counter_collection.csv > $OUT/sq_c3_lstm.jsonl; cat $OUT/sq_c3_lstm.jsonl
tail -3 $OUT/pmc_c3_3.log
