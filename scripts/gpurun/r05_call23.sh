#!/bin/bash
# Round 5, twenty-third GPU call: EV districts with the load hint back on their instantiation; headline profile (kernel stats + counters) on this tree.
set -u
TAG=r05x
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python scripts/ev_step_bench.py > $OUT/ev_step_bench.log 2>$OUT/ev_step_bench.err; tail -1 $OUT/ev_step_bench.log | cut -c1-600
kernel_of() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['roofline']['kernel'].split('+')[int(sys.argv[2])])" "$1" "${2:-0}"; }
pmc_pass() { local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $OUT/pmc_$name -o run -- "$@" > /dev/null 2>$OUT/pmc_$name.log; }
trace() { local name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o run -- "$@" > $OUT/under_rocprof_$name.json 2>$OUT/trace_$name.log
  cp $OUT/trace_$name/*kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null; }
counters() {
  local name=$1 kern=$2; shift 3
  for ctr in FETCH_SIZE WRITE_SIZE; do pmc_pass ${name}_$ctr $ctr -- "$@"; done
  python scripts/pmc_summary.py $OUT/${TAG}_${name}_pmc_summary.json "$kern" $OUT/pmc_${name}_FETCH_SIZE/*counter_collection.csv $OUT/pmc_${name}_WRITE_SIZE/*counter_collection.csv > /dev/null; }
FAIL=0
chk() { python scripts/check_profiles.py "$@" >> $OUT/check.log || FAIL=1; }
python bench.py > $OUT/bench_line.json 2>$OUT/bench_line.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_driver_flags.json 2>/dev/null
BENCH="python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-streaming --no-traffic-pass --no-chain-entry"
trace bench $BENCH
K=$(kernel_of $OUT/bench_line.json)
counters bench "$K" -- python bench.py --steps 300 --warmup 100 --no-cpu-baseline --no-graph --no-streaming --no-traffic-pass --no-chain-entry
chk --duration-tol 0.05 $OUT/bench_line.json $OUT/${TAG}_bench_pmc_summary.json $OUT/bench_kernel_stats.csv
for c in C4-lean; do
  python bench.py --config $c --reps 1 --steps 500 > $OUT/tmp.json 2>/dev/null
  counters c4lean "$(kernel_of $OUT/tmp.json)" -- python bench.py --config $c --steps 300 --warmup 50 --reps 1 --no-graph
  python bench.py --config $c --traffic-summary $OUT/${TAG}_c4lean_pmc_summary.json > $OUT/bench_$c.json 2>$OUT/bench_$c.err
  trace c4lean python bench.py --config $c --reps 1
  chk $OUT/bench_$c.json $OUT/${TAG}_c4lean_pmc_summary.json $OUT/c4lean_kernel_stats.csv
done
python bench.py --config C4-lean --envs-per-gpu 8192 > $OUT/bench_C4-lean_8192.json 2>/dev/null
cat $OUT/check.log
python - <<PY
import json, glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], 'value %.3e' % d['value'], 'ms_per_step', d['ms_per_step'], 'launch_us %.2f' % r['launch_us'], 'frac %.3f' % r['frac'], r['kernel'], 'traffic', r.get('traffic'))
PY
grep -h "cl_step_lean_kernel\|cl_step_lean_chunk" $OUT/bench_kernel_stats.csv $OUT/c4lean_kernel_stats.csv | cut -c1-200
rm -rf $OUT/pmc_* 
echo FAIL=$FAIL
