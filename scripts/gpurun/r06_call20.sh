#!/bin/bash
# Round 6, twentieth GPU call: SQ counters of the battery + PV fused rollout (C5: 17 x 32 768, 24 steps per launch) and of C4-lean-B, both precision models.
set -u
OUT=gpurun_out/r06t; mkdir -p $OUT; export TMPDIR=/tmp
CTR1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
CTR2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVES"
for c in C5 C4-lean-B; do
for p in chain fp32; do
  python bench.py --config $c --precision $p --reps 3 > $OUT/${c}_$p.json 2>/dev/null
  n=1
  for C in "$CTR1" "$CTR2"; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${c}_${p}_$n -o run -- python bench.py --config $c --precision $p --steps 12 --warmup 3 --reps 1 --no-graph > /dev/null 2>$OUT/pmc_${c}_${p}_$n.log
    n=$((n+1))
  done
  python scripts/pmc_by_kernel.py cl_rollout $OUT/pmc_${c}_${p}_1/*counter_collection.csv $OUT/pmc_${c}_${p}_2/*counter_collection.csv > $OUT/sq_${c}_$p.jsonl
  echo "== $c $p"; cat $OUT/sq_${c}_$p.jsonl
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06t/*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], 'frac', r['frac'], r['kernel'])
PY
