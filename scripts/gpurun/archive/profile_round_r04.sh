#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun): bench lines (default and the driver's --steps 20 --warmup 5), rocprofv3
# kernel stats of the same command, HBM traffic counters in separate passes (collected on the kernel the bench line names:
# scripts/check_profiles.py fails the run otherwise), one line + counters per BASELINE config, the A-kpi line, user-level step timings.
# Outputs under gpurun_out/prof_$TAG/; copy what should be judged into profiles/ (scripts/collect_profiles.sh).
set -u
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
kernel_of() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['roofline']['kernel'].split('+')[int(sys.argv[2])])" "$1" "${2:-0}"; }
pmc_pass() {   # pmc_pass <name> <counters...> -- <command...>: one rocprofv3 --pmc run (no tracing domains beside it)
  local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $OUT/pmc_$name -o run -- "$@" > /dev/null 2>$OUT/pmc_$name.log
}
FAIL=0
# ---- headline ----
python bench.py > $OUT/bench_line.json 2>$OUT/bench_line.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_line_driver_flags.json 2>$OUT/bench_line_driver_flags.err
BENCH="python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-streaming --no-traffic-pass"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2>$OUT/trace.log
cp $OUT/trace/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
K=$(kernel_of $OUT/bench_line.json)
for c in FETCH_SIZE WRITE_SIZE; do
  pmc_pass bench_$c $c -- python bench.py --steps 300 --warmup 100 --no-cpu-baseline --no-graph --no-streaming --no-traffic-pass
done
python scripts/pmc_summary.py $OUT/bench_pmc_summary.json "$K" $OUT/pmc_bench_FETCH_SIZE/*counter_collection.csv $OUT/pmc_bench_WRITE_SIZE/*counter_collection.csv > /dev/null
python scripts/check_profiles.py --duration-tol 0.05 $OUT/bench_line.json $OUT/bench_pmc_summary.json $OUT/bench_kernel_stats.csv >> $OUT/check.log || FAIL=1
# ---- HBM-streaming entry (17 x 1 048 576): counters on the kernel that line names ----
python bench.py --envs-per-gpu 1048576 --steps 20 --warmup 5 --reps 3 --no-cpu-baseline > $OUT/bench_streaming_line.json 2>$OUT/bench_streaming_line.err
KS=$(kernel_of $OUT/bench_streaming_line.json)
for c in FETCH_SIZE WRITE_SIZE; do
  pmc_pass streaming_$c $c -- python bench.py --envs-per-gpu 1048576 --steps 30 --warmup 5 --reps 1 --no-cpu-baseline --no-graph
done
python scripts/pmc_summary.py $OUT/streaming_pmc_summary.json "$KS" $OUT/pmc_streaming_FETCH_SIZE/*counter_collection.csv $OUT/pmc_streaming_WRITE_SIZE/*counter_collection.csv > /dev/null
python scripts/check_profiles.py $OUT/bench_streaming_line.json $OUT/streaming_pmc_summary.json >> $OUT/check.log || FAIL=1
# rocprofv3 duration of the HBM-true shape (the judge's round-3 gap): same command shape as the line above; the kernel's AverageNs must
# agree with the line's HIP-event launch_us within 5 %
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_streaming -o run -- python bench.py --envs-per-gpu 1048576 --steps 200 --warmup 20 --reps 3 --no-cpu-baseline > $OUT/bench_streaming_under_rocprof.json 2>$OUT/trace_streaming.log
cp $OUT/trace_streaming/*kernel_stats.csv $OUT/streaming_kernel_stats.csv 2>/dev/null
python scripts/check_profiles.py --duration-tol 0.05 $OUT/bench_streaming_line.json $OUT/streaming_kernel_stats.csv >> $OUT/check.log || FAIL=1
# ---- one line per BASELINE config (+ HBM counters for the A-mode ones, kernel stats for all) ----
for c in C2 C3 C4 C4-lean C5 T9; do
  n=$(echo $c | tr 'A-Z' 'a-z' | tr -d '-')
  python bench.py --config $c > $OUT/bench_$c.json 2>$OUT/bench_$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$n -o run -- python bench.py --config $c --reps 1 > /dev/null 2>$OUT/trace_$n.log
  cp $OUT/trace_$n/*kernel_stats.csv $OUT/${n}_kernel_stats.csv 2>/dev/null
  python scripts/check_profiles.py $OUT/bench_$c.json $OUT/${n}_kernel_stats.csv >> $OUT/check.log || FAIL=1
  case $c in C2|C4|C4-lean|T9)
    KC=$(kernel_of $OUT/bench_$c.json)
    for ctr in FETCH_SIZE WRITE_SIZE; do
      pmc_pass ${n}_$ctr $ctr -- python bench.py --config $c --steps 300 --warmup 50 --reps 1 --no-graph
    done
    python scripts/pmc_summary.py $OUT/${n}_pmc_summary.json "$KC" $OUT/pmc_${n}_FETCH_SIZE/*counter_collection.csv $OUT/pmc_${n}_WRITE_SIZE/*counter_collection.csv > /dev/null
    python scripts/check_profiles.py $OUT/bench_$c.json $OUT/${n}_pmc_summary.json >> $OUT/check.log || FAIL=1;;
  esac
done
# C4 shards with the second launch per step (cl_tuning.finish = 1) next to the deferred finish of the lines above: same box, same session
for c in C4 C4-lean; do
  CL_TUNE_FINISH=1 python bench.py --config $c > $OUT/bench_${c}_second_launch.json 2>/dev/null
done
# C3 / C5: vector-ALU counters of the LSTM and the rollout kernel (their bound is instruction issue, not HBM)
pmc_pass c3_SQ SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -- python bench.py --config C3 --steps 60 --warmup 20 --reps 1 --no-graph
python scripts/pmc_by_kernel.py cl_lstm_kernel $OUT/pmc_c3_SQ/*counter_collection.csv > $OUT/c3_lstm_sq_by_kernel.jsonl
pmc_pass c5_SQ SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -- python bench.py --config C5 --steps 40 --warmup 10 --reps 1 --no-graph
python scripts/pmc_by_kernel.py cl_rollout_kernel $OUT/pmc_c5_SQ/*counter_collection.csv > $OUT/c5_rollout_sq_by_kernel.jsonl
# ---- streaming KPIs (mode A-kpi), CLD_F64_MAPS cost, user-level step ----
python bench.py --kpi --no-streaming --no-cpu-baseline --no-traffic-pass > $OUT/bench_kpi.json 2>$OUT/bench_kpi.err
python bench.py --kpi --config C3 > $OUT/bench_kpi_C3.json 2>$OUT/bench_kpi_C3.err
# thermal district with streaming KPIs inside the step launch (cl_step_full_kpi_kernel): line, kernel stats, HBM counters
python bench.py --kpi --config T9 --no-cpu-baseline > $OUT/bench_kpi_T9.json 2>$OUT/bench_kpi_T9.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_kpi_t9 -o run -- python bench.py --kpi --config T9 --reps 1 --no-cpu-baseline > /dev/null 2>$OUT/trace_kpi_t9.log
cp $OUT/trace_kpi_t9/*kernel_stats.csv $OUT/kpi_t9_kernel_stats.csv 2>/dev/null
python scripts/check_profiles.py $OUT/bench_kpi_T9.json $OUT/kpi_t9_kernel_stats.csv >> $OUT/check.log || FAIL=1
KC=$(kernel_of $OUT/bench_kpi_T9.json)
for ctr in FETCH_SIZE WRITE_SIZE; do
  pmc_pass kpi_t9_$ctr $ctr -- python bench.py --kpi --config T9 --steps 300 --warmup 50 --reps 1 --no-graph --no-cpu-baseline
done
python scripts/pmc_summary.py $OUT/kpi_t9_pmc_summary.json "$KC" $OUT/pmc_kpi_t9_FETCH_SIZE/*counter_collection.csv $OUT/pmc_kpi_t9_WRITE_SIZE/*counter_collection.csv > /dev/null
python scripts/check_profiles.py $OUT/bench_kpi_T9.json $OUT/kpi_t9_pmc_summary.json >> $OUT/check.log || FAIL=1
python bench.py --f64-maps --no-streaming --no-cpu-baseline --no-traffic-pass > $OUT/bench_f64_maps.json 2>$OUT/bench_f64_maps.err
for s in env_step_bench f64_cost ev_step_bench observe_bench; do
  timeout 600 python scripts/$s.py > $OUT/${s}.log 2>$OUT/$s.err
done
cat $OUT/check.log
echo "profile check: FAIL=$FAIL"
ls $OUT | head -80
exit $FAIL
