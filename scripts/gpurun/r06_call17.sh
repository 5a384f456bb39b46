#!/bin/bash
# Round 6, seventeenth GPU call: the chunk rules after the sweep -- re-read the selection map, re-measure the sweep's default column, tests.
set -u
OUT=gpurun_out/r06q; mkdir -p $OUT; export TMPDIR=/tmp
python scripts/r06_selection_map.py $OUT/kernel_selection_r06.json 2>$OUT/map.err | tail -30
rm -f $OUT/chunks_*.jsonl
timeout 600 python scripts/r06_chunk_sweep.py $OUT/chunks_chain.jsonl chain 2>$OUT/sweep_chain.err | cut -c1-260
timeout 600 python scripts/r06_chunk_sweep.py $OUT/chunks_fp32.jsonl fp32 2>$OUT/sweep_fp32.err | cut -c1-260
cp $OUT/kernel_selection_r06.json tests/golden/kernel_selection_r06.json
timeout 1200 python -m pytest tests/test_gpu_config_sizes.py tests/test_gpu_rollout.py tests/test_gpu_parity.py -m gpu -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail
for p in chain fp32; do python bench.py --config C4 --precision $p > $OUT/C4_$p.json 2>/dev/null; python bench.py --config C4 --precision $p --envs-per-gpu 8192 --steps 2000 --reps 3 > $OUT/C4_${p}_8192.json 2>/dev/null; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06q/C4*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'value %.3e' % d['value'], 'ms/step %.5f' % d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], 'frac', r['frac'], r['kernel'])
PY
