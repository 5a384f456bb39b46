#!/bin/bash
# Round 6, twenty-third GPU call: SQ counters of the packed thermal rollout as it stands (actions cached in LDS, 8-wave workgroups) -- the counts bench.py's VALU roofline uses.
set -u
OUT=gpurun_out/r06w; mkdir -p $OUT; export TMPDIR=/tmp
CTR1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
CTR2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVES"
for p in chain fp32; do
  n=1
  for C in "$CTR1" "$CTR2"; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${p}_$n -o run -- python bench.py --config C4-B --precision $p --steps 12 --warmup 3 --reps 1 --no-graph > /dev/null 2>$OUT/pmc_${p}_$n.log
    n=$((n+1))
  done
  python scripts/pmc_by_kernel.py cl_rollout $OUT/pmc_${p}_1/*counter_collection.csv $OUT/pmc_${p}_2/*counter_collection.csv > $OUT/sq_c4b_$p.jsonl; cat $OUT/sq_c4b_$p.jsonl
done
