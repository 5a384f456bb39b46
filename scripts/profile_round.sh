#!/bin/bash
# Round profile recipe (GPU box, through gpurun; the round-4 recipe: scripts/gpurun/profile_round_r04.sh): the default bench line and the driver's flags, rocprofv3 kernel stats of the same commands,
# HBM counters in separate passes (collected on the kernel the line names; check_profiles.py also holds a line to the traffic of the summary it
# cites), the HBM-streaming shape, one line (+ stats, + counters where HBM-bound) per BASELINE config, BASELINE config 4 WHOLE on one GPU in
# mode A and in mode B, the CLD_F64_CHAIN lines, the KPI / float64-reference lines, user-level timings, the GPU suite.
# Outputs under gpurun_out/prof_$TAG/; copy what should be judged into profiles/.
set -u
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
kernel_of() { python -c "import json,sys; r=json.load(open(sys.argv[1]))['roofline']; r=r.get('metric_shape', r) if len(sys.argv) > 3 else r; print(r['kernel'].split('+')[int(sys.argv[2])])" "$1" "${2:-0}" ${3:-}; }
pmc_pass() { local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $OUT/pmc_$name -o run -- "$@" > /dev/null 2>$OUT/pmc_$name.log; }
trace() { local name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o run -- "$@" > $OUT/under_rocprof_$name.json 2>$OUT/trace_$name.log
  cp $OUT/trace_$name/*kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null; }
counters() {   # counters <name> <kernel> -- <command...>: FETCH_SIZE / WRITE_SIZE passes + summary
  local name=$1 kern=$2; shift 3
  for ctr in FETCH_SIZE WRITE_SIZE; do pmc_pass ${name}_$ctr $ctr -- "$@"; done
  python scripts/pmc_summary.py $OUT/${TAG}_${name}_pmc_summary.json "$kern" $OUT/pmc_${name}_FETCH_SIZE/*counter_collection.csv $OUT/pmc_${name}_WRITE_SIZE/*counter_collection.csv > /dev/null; }
FAIL=0
chk() { python scripts/check_profiles.py "$@" >> $OUT/check.log || FAIL=1; }
# ---- GPU suite ----
(timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_suite.log 2>&1; echo "rc=$?" >> $OUT/gpu_suite.log); tail -4 $OUT/gpu_suite.log
# ---- headline (default precision model = CLD_F64_CHAIN).  The line's `roofline` is the HBM-true 17 x 1 048 576 launch, `roofline.metric_shape` the
# ---- cache-resident 17 x 65 536 launch `value` is timed on: rocprofv3 stats + counters for BOTH, each checked against its entry of the line
python bench.py > $OUT/bench_line.json 2>$OUT/bench_line.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_line_driver_flags.json 2>/dev/null
BENCH="python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-streaming --no-traffic-pass --no-side-entries"
trace bench $BENCH
K=$(kernel_of $OUT/bench_line.json 0 metric)
counters bench "$K" -- python bench.py --steps 300 --warmup 100 --no-cpu-baseline --no-graph --no-streaming --no-traffic-pass --no-side-entries
chk --duration-tol 0.05 --entry metric_shape $OUT/bench_line.json $OUT/${TAG}_bench_pmc_summary.json $OUT/bench_kernel_stats.csv
# ---- HBM-streaming shape (17 x 1 048 576): the line's top-level roofline ----
SB="python bench.py --envs-per-gpu 1048576 --no-cpu-baseline --no-streaming --no-traffic-pass --no-side-entries"
KS=$(kernel_of $OUT/bench_line.json)
counters streaming "$KS" -- $SB --steps 30 --warmup 5 --reps 1 --no-graph
$SB --steps 20 --warmup 5 --reps 3 --traffic-summary $OUT/${TAG}_streaming_pmc_summary.json > $OUT/bench_streaming_line.json 2>/dev/null
trace streaming $SB --steps 200 --warmup 20 --reps 3
chk --duration-tol 0.05 $OUT/bench_streaming_line.json $OUT/${TAG}_streaming_pmc_summary.json $OUT/streaming_kernel_stats.csv
chk --duration-tol 0.05 $OUT/bench_line.json $OUT/streaming_kernel_stats.csv
# ---- one line per BASELINE config ----
for c in C2 C3 C4 C4-lean C5 T9; do
  n=$(echo $c | tr 'A-Z' 'a-z' | tr -d '-')
  case $c in C2|C4|C4-lean|T9)
    python bench.py --config $c --reps 1 --steps 500 > $OUT/tmp.json 2>/dev/null
    counters $n "$(kernel_of $OUT/tmp.json)" -- python bench.py --config $c --steps 300 --warmup 50 --reps 1 --no-graph
    python bench.py --config $c --traffic-summary $OUT/${TAG}_${n}_pmc_summary.json > $OUT/bench_$c.json 2>$OUT/bench_$c.err
    chk $OUT/bench_$c.json $OUT/${TAG}_${n}_pmc_summary.json;;
  *) python bench.py --config $c > $OUT/bench_$c.json 2>$OUT/bench_$c.err;;
  esac
  trace $n python bench.py --config $c --reps 1
  chk $OUT/bench_$c.json $OUT/${n}_kernel_stats.csv
done
# ---- BASELINE config 4 whole on one GPU: mode A with counters, mode B ----
for c in C4 C4-lean; do
  n=$(echo $c | tr 'A-Z' 'a-z' | tr -d '-')_8192
  python bench.py --config $c --envs-per-gpu 8192 --steps 500 --reps 1 > $OUT/tmp.json 2>/dev/null
  counters $n "$(kernel_of $OUT/tmp.json)" -- python bench.py --config $c --envs-per-gpu 8192 --steps 200 --warmup 40 --reps 1 --no-graph
  python bench.py --config $c --envs-per-gpu 8192 --steps 2000 --reps 3 --traffic-summary $OUT/${TAG}_${n}_pmc_summary.json > $OUT/bench_${c}_8192.json 2>/dev/null
  trace $n python bench.py --config $c --envs-per-gpu 8192 --steps 2000 --reps 1
  chk --duration-tol 0.05 $OUT/bench_${c}_8192.json $OUT/${TAG}_${n}_pmc_summary.json $OUT/${n}_kernel_stats.csv
done
for c in C4-B C4-lean-B; do
  for E in 1024 8192; do python bench.py --config $c --envs-per-gpu $E > $OUT/bench_${c}_$E.json 2>/dev/null; done
done
trace c4leanb python bench.py --config C4-lean-B --reps 1
trace c4b python bench.py --config C4-B --reps 1
chk $OUT/bench_C4-B_1024.json $OUT/c4b_kernel_stats.csv
for E in 1024 8192; do python bench.py --config C4-B --envs-per-gpu $E --precision fp32 > $OUT/bench_fp32_C4-B_$E.json 2>/dev/null; done
# ---- the all-fp32 map (the opt-in throughput mode) and the float64 reference mode ----
python bench.py --precision fp32 --no-cpu-baseline --no-traffic-pass > $OUT/bench_fp32.json 2>$OUT/bench_fp32.err
trace fp32 $BENCH --precision fp32
chk --duration-tol 0.05 --entry metric_shape $OUT/bench_fp32.json $OUT/fp32_kernel_stats.csv
for c in T9 C4 C4-lean C5; do python bench.py --config $c --precision fp32 > $OUT/bench_fp32_$c.json 2>/dev/null; done
python bench.py --precision f64 --no-streaming --no-cpu-baseline --no-traffic-pass > $OUT/bench_f64_maps.json 2>/dev/null
# ---- streaming KPIs ----
python bench.py --kpi --no-streaming --no-cpu-baseline --no-traffic-pass > $OUT/bench_kpi.json 2>/dev/null
python bench.py --kpi --config T9 --no-cpu-baseline > $OUT/bench_kpi_T9.json 2>/dev/null
python bench.py --kpi --config T9 --precision fp32 --no-cpu-baseline > $OUT/bench_kpi_T9_fp32.json 2>/dev/null
# ---- user-level timings ----
for s in env_step_bench observe_bench ev_step_bench; do timeout 400 python scripts/$s.py > $OUT/${s}.log 2>$OUT/$s.err; done
cat $OUT/check.log
echo "profile check: FAIL=$FAIL"
python - <<PY
import json, glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], 'value %.3e' % d['value'], 'launch_us %.2f' % r['launch_us'], 'frac', r['frac'], r['kernel'], 'traffic', r.get('traffic'))
        if 'metric_shape' in r:
            m = r['metric_shape']; print('    metric_shape: launch_us %.3f' % m['launch_us'], 'frac %.3f' % m['frac'], m['kernel'], 'traffic', m.get('traffic'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
exit $FAIL
