#!/bin/bash
# Round 6, thirteenth GPU call: workgroup size of the chunked packed thermal rollout (waves = buildings per chunk) x batch size x precision,
# and SQ counters of the mode-A thermal kernels (C4 shard, T9) for the same per-wave accounting.
set -u
OUT=gpurun_out/r06m; mkdir -p $OUT; export TMPDIR=/tmp
for p in chain fp32; do
  for E in 1024 2048 8192; do
    for nw in 4 8 16; do
      CL_TUNE_B_CHUNK=$nw CL_TUNE_NW=$nw python bench.py --config C4-B --precision $p --envs-per-gpu $E --reps 3 > $OUT/C4-B_${p}_${E}_nw$nw.json 2>$OUT/err.log
    done
  done
done
CTR1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
CTR2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES"
for c in C4 T9; do
  for p in chain fp32; do
    n=1
    for C in "$CTR1" "$CTR2"; do
      rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${c}_${p}_$n -o run -- python bench.py --config $c --precision $p --steps 50 --warmup 10 --reps 1 --no-graph > /dev/null 2>$OUT/pmc_${c}_${p}_$n.log
      n=$((n+1))
    done
    python scripts/pmc_by_kernel.py cl_step $OUT/pmc_${c}_${p}_1/*counter_collection.csv $OUT/pmc_${c}_${p}_2/*counter_collection.csv > $OUT/sq_${c}_${p}.jsonl
    echo "== $c $p"; cat $OUT/sq_${c}_${p}.jsonl
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06m/*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], r['kernel'])
    except Exception as e: print(f, 'unreadable', e)
PY
