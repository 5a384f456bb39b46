#!/bin/bash
# Round 5, first GPU call: the GPU suite on the new code, the default bench line (reference as the stated CPU baseline, full-year table),
# the 720-hour table beside it, env-major variants, BASELINE config 4 WHOLE on one GPU (mode A with counters, mode B).
set -u
OUT=gpurun_out/r05a; mkdir -p $OUT; export TMPDIR=/tmp
kernel_of() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['roofline']['kernel'].split('+')[int(sys.argv[2])])" "$1" "${2:-0}"; }
pmc_pass() { local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $OUT/pmc_$name -o run -- "$@" > /dev/null 2>$OUT/pmc_$name.log; }
(timeout 1000 python -m pytest tests -m gpu -q --maxfail=12 > $OUT/gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/gpu_tests.log)
tail -25 $OUT/gpu_tests.log
python bench.py > $OUT/bench_line.json 2>$OUT/bench_line.err; tail -c 600 $OUT/bench_line.err
python bench.py --table-hours 720 --no-cpu-baseline --no-traffic-pass > $OUT/bench_line_720h.json 2>/dev/null
timeout 400 python scripts/r05_envmajor_ab.py > $OUT/envmajor_ab.log 2>&1; cat $OUT/envmajor_ab.log
for c in C4 C4-lean; do
  n=$(echo $c | tr 'A-Z' 'a-z' | tr -d '-')
  for E in 1024 8192; do
    python bench.py --config $c --envs-per-gpu $E --steps 2000 --reps 3 > $OUT/tmp_line.json 2>$OUT/bench_${c}_$E.err
    KC=$(kernel_of $OUT/tmp_line.json)
    for ctr in FETCH_SIZE WRITE_SIZE; do
      pmc_pass ${n}_${E}_$ctr $ctr -- python bench.py --config $c --envs-per-gpu $E --steps 200 --warmup 40 --reps 1 --no-graph
    done
    python scripts/pmc_summary.py $OUT/r05_${n}_${E}_pmc_summary.json "$KC" $OUT/pmc_${n}_${E}_FETCH_SIZE/*counter_collection.csv $OUT/pmc_${n}_${E}_WRITE_SIZE/*counter_collection.csv > /dev/null
    python bench.py --config $c --envs-per-gpu $E --steps 2000 --reps 3 --traffic-summary $OUT/r05_${n}_${E}_pmc_summary.json > $OUT/bench_${c}_$E.json 2>>$OUT/bench_${c}_$E.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_${n}_$E -o run -- python bench.py --config $c --envs-per-gpu $E --steps 2000 --reps 1 > /dev/null 2>$OUT/trace_${n}_$E.log
    cp $OUT/trace_${n}_$E/*kernel_stats.csv $OUT/r05_${n}_${E}_kernel_stats.csv 2>/dev/null
    python scripts/check_profiles.py --duration-tol 0.05 $OUT/bench_${c}_$E.json $OUT/r05_${n}_${E}_pmc_summary.json $OUT/r05_${n}_${E}_kernel_stats.csv >> $OUT/check.log
    python -c "import json;d=json.load(open('$OUT/bench_${c}_$E.json'));r=d['roofline'];print('$c',$E,'value %.3e'%d['value'],'launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],'traffic',r.get('traffic'),'alg',r['algorithmic_bytes_per_unit']*r['units_per_launch'],r['kernel'])"
  done
done
for c in C4-B C4-lean-B; do
  for E in 1024 8192; do
    python bench.py --config $c --envs-per-gpu $E > $OUT/bench_${c}_$E.json 2>$OUT/bench_${c}_$E.err
    python -c "import json;d=json.load(open('$OUT/bench_${c}_$E.json'));r=d['roofline'];print('$c',$E,'value %.3e'%d['value'],'launch_us %.2f'%r['launch_us'],'per step %.2f us'%(r['launch_us']/24),'frac %.3f'%r['frac'],r['kernel'])"
  done
done
cat $OUT/check.log
python -c "import json;d=json.load(open('$OUT/bench_line.json'));print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['hbm_streaming']['launch_us'], d['roofline']['hbm_streaming']['frac'], d['roofline']['hbm_streaming']['kernel']); c=d['cpu_baseline']; print(c['kind'], c['value'], c['cores'], c.get('c1_single_process'), c['port']['value'])"
python -c "import json;d=json.load(open('$OUT/bench_line_720h.json'));print('720h', {k:d[k] for k in ('value','ms_per_step')}, d['roofline']['launch_us'], d['roofline']['hbm_streaming']['launch_us'])"
