#!/bin/bash
set -u
OUT=gpurun_out/r06z6; mkdir -p $OUT; export TMPDIR=/tmp
CTR1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for p in chain fp32; do
  rocprofv3 --pmc $CTR1 --output-format csv -d $OUT/pmc_${p} -o run -- python bench.py --config C4-B --envs-per-gpu 8192 --precision $p --steps 6 --warmup 2 --reps 1 --no-graph > /dev/null 2>$OUT/pmc_${p}.log
  python scripts/pmc_by_kernel.py cl_rollout $OUT/pmc_${p}/*counter_collection.csv > $OUT/sq_c4b_8192_$p.jsonl; cat $OUT/sq_c4b_8192_$p.jsonl
done
