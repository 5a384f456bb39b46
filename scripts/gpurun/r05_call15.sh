#!/bin/bash
# Round 5, fifteenth GPU call: BASELINE config 4 whole on one GPU after the chunk-geometry change -- bench line, rocprofv3 kernel stats, HBM counters
# (separate passes), check; the 4096-env half; the full GPU suite on this tree.
set -u
TAG=r05p
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
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
for E in 8192 4096; do
  c=C4; n=c4_$E
  python bench.py --config $c --envs-per-gpu $E --steps 500 --reps 1 > $OUT/tmp.json 2>/dev/null
  counters $n "$(kernel_of $OUT/tmp.json)" -- python bench.py --config $c --envs-per-gpu $E --steps 200 --warmup 40 --reps 1 --no-graph
  python bench.py --config $c --envs-per-gpu $E --steps 2000 --reps 3 --traffic-summary $OUT/${TAG}_${n}_pmc_summary.json > $OUT/bench_${c}_$E.json 2>/dev/null
  trace $n python bench.py --config $c --envs-per-gpu $E --steps 2000 --reps 1
  chk --duration-tol 0.05 $OUT/bench_${c}_$E.json $OUT/${TAG}_${n}_pmc_summary.json $OUT/${n}_kernel_stats.csv
  python -c "
import json
d=json.load(open('$OUT/bench_${c}_$E.json')); r=d['roofline']
print('$c', $E, 'launch_us %.2f'%r['launch_us'], 'frac %.3f'%r['frac'], 'traffic', r['traffic'], r['kernel'])
"
  grep -i "cl_step_full" $OUT/${n}_kernel_stats.csv | head -2
done
python bench.py --config C4 > $OUT/bench_C4.json 2>/dev/null
python bench.py --config C4-lean --envs-per-gpu 8192 > $OUT/bench_C4-lean_8192.json 2>/dev/null
cat $OUT/check.log
rm -rf $OUT/pmc_* $OUT/trace_*/*.db 2>/dev/null
(timeout 1300 python -m pytest tests -m gpu -q --maxfail=20 > $OUT/gpu_suite.log 2>&1; echo "rc=$?" >> $OUT/gpu_suite.log); tail -5 $OUT/gpu_suite.log
echo FAIL=$FAIL
