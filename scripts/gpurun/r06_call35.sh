#!/bin/bash
# Round 6, thirty-fifth GPU call: the bench lines that changed after the profile session (call 26): C4-lean-B at two envs per lane under the chain, T9 with its mode-B entry, C4-B.
set -u
OUT=gpurun_out/r06z5; mkdir -p $OUT; export TMPDIR=/tmp
for E in 1024 8192; do
  python bench.py --config C4-lean-B --envs-per-gpu $E > $OUT/bench_C4-lean-B_$E.json 2>/dev/null
  python bench.py --config C4-B --envs-per-gpu $E > $OUT/bench_C4-B_$E.json 2>/dev/null
done
python bench.py --config T9 > $OUT/bench_T9.json 2>/dev/null
python bench.py --config T9 --precision fp32 > $OUT/bench_fp32_T9.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c4leanb -o run -- python bench.py --config C4-lean-B --reps 1 > $OUT/under_rocprof_c4leanb.json 2>/dev/null
cp $OUT/trace_c4leanb/*kernel_stats.csv $OUT/c4leanb_kernel_stats.csv
python scripts/check_profiles.py $OUT/bench_C4-lean-B_1024.json $OUT/c4leanb_kernel_stats.csv
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06z5/bench_*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], 'frac', r['frac'], r['kernel'], (r.get('mode_b') or {}).get('us_per_step'))
PY
