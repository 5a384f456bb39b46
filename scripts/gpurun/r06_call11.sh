#!/bin/bash
# Round 6, eleventh GPU call: SQ counters of the packed thermal rollout before (HEAD build: blocks drawn per step) and after (blocks cached in LDS).
set -u
OUT=gpurun_out/r06k; mkdir -p $OUT; export TMPDIR=/tmp
CTR1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
CTR2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM"
for v in default head; do
  lib=citylearn_amd/libcitylearn_amd.so; [ $v != default ] && lib=citylearn_amd/libcitylearn_amd_$v.so
  for p in chain fp32; do
    CITYLEARN_AMD_LIB=$lib python bench.py --config C4-B --precision $p --reps 3 > $OUT/C4-B_${p}_$v.json 2>/dev/null
    n=1
    for C in "$CTR1" "$CTR2"; do
      CITYLEARN_AMD_LIB=$lib rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${p}_${v}_$n -o run -- python bench.py --config C4-B --precision $p --steps 12 --warmup 3 --reps 1 --no-graph > /dev/null 2>$OUT/pmc_${p}_${v}_$n.log
      n=$((n+1))
    done
    python scripts/pmc_by_kernel.py cl_rollout $OUT/pmc_${p}_${v}_1/*counter_collection.csv $OUT/pmc_${p}_${v}_2/*counter_collection.csv > $OUT/sq_${p}_$v.jsonl
    echo "== $v $p"; cat $OUT/sq_${p}_$v.jsonl
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06k/*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
PY
