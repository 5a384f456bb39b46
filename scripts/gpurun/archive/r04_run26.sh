#!/bin/bash
# C4 thermal shard re-profiled after the early-first-building change (same recipe as scripts/profile_round.sh, C4 rows only)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_r04b; mkdir -p $OUT
kernel_of() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['roofline']['kernel'].split('+')[0])" "$1"; }
for c in C4 C4-lean; do
  n=$(echo $c | tr 'A-Z' 'a-z' | tr -d '-')
  python bench.py --config $c > $OUT/bench_$c.json 2>/dev/null
  CL_TUNE_FINISH=1 python bench.py --config $c > $OUT/bench_${c}_second_launch.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$n -o run -- python bench.py --config $c --reps 1 > /dev/null 2>$OUT/trace_$n.log
  cp $OUT/trace_$n/*kernel_stats.csv $OUT/${n}_kernel_stats.csv
  KC=$(kernel_of $OUT/bench_$c.json)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${n}_$ctr -o run -- python bench.py --config $c --steps 300 --warmup 50 --reps 1 --no-graph > /dev/null 2>&1
  done
  python scripts/pmc_summary.py $OUT/${n}_pmc_summary.json "$KC" $OUT/pmc_${n}_FETCH_SIZE/*counter_collection.csv $OUT/pmc_${n}_WRITE_SIZE/*counter_collection.csv > /dev/null
  python scripts/check_profiles.py --duration-tol 0.05 $OUT/bench_$c.json $OUT/${n}_kernel_stats.csv $OUT/${n}_pmc_summary.json
  python -c "
import json
for f in ('bench_$c.json','bench_${c}_second_launch.json'):
    d=json.load(open('$OUT/'+f)); r=d['roofline']; print(f, '%.2f us frac %.3f' % (r['launch_us'], r['frac']), r['kernel'])"
done
