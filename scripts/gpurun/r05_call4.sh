#!/bin/bash
# Round 5, fourth GPU call: GPU suite after the swapped (chunk-major) grids; C4 shard counters; env-major with direct-to-LDS action loads.
set -u
OUT=gpurun_out/r05d; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 > $OUT/gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/gpu_tests.log)
tail -30 $OUT/gpu_tests.log
kernel_of() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['roofline']['kernel'].split('+')[int(sys.argv[2])])" "$1" "${2:-0}"; }
pmc_pass() { local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $OUT/pmc_$name -o run -- "$@" > /dev/null 2>$OUT/pmc_$name.log; }
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],'value %.3e'%d['value'],'launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],r['kernel'],'traffic',r.get('traffic'),'alg',r.get('algorithmic_bytes_per_unit',0)*r['units_per_launch'])" "$1" "$2"; }
S="--no-cpu-baseline --no-traffic-pass --no-streaming --steps 20 --warmup 5 --reps 3 --envs-per-gpu 1048576"
for i in 1 2 3; do
  python bench.py $S > $OUT/s_base_$i.json 2>/dev/null; show $OUT/s_base_$i.json "streaming base"
  CL_TUNE_LEAN_VARIANT=16 python bench.py $S > $OUT/s_adma_$i.json 2>$OUT/s_adma.err; show $OUT/s_adma_$i.json "streaming ADMA"
done
CL_TUNE_LEAN_VARIANT=16 python bench.py --no-cpu-baseline --no-traffic-pass --no-streaming --steps 20 --warmup 5 --reps 3 --envs-per-gpu 262144 > $OUT/s18_adma.json 2>/dev/null; show $OUT/s18_adma.json "2^18 ADMA"
python bench.py --no-cpu-baseline --no-traffic-pass --no-streaming --steps 20 --warmup 5 --reps 3 --envs-per-gpu 262144 > $OUT/s18_base.json 2>/dev/null; show $OUT/s18_base.json "2^18 base"
for c in C4 C4-lean; do
  n=$(echo $c | tr 'A-Z' 'a-z' | tr -d '-')
  python bench.py --config $c --steps 2000 --reps 3 > $OUT/tmp_line.json 2>$OUT/bench_${c}.err
  KC=$(kernel_of $OUT/tmp_line.json)
  for ctr in FETCH_SIZE WRITE_SIZE; do pmc_pass ${n}_$ctr $ctr -- python bench.py --config $c --steps 200 --warmup 40 --reps 1 --no-graph; done
  python scripts/pmc_summary.py $OUT/r05d_${n}_pmc_summary.json "$KC" $OUT/pmc_${n}_FETCH_SIZE/*counter_collection.csv $OUT/pmc_${n}_WRITE_SIZE/*counter_collection.csv > /dev/null
  python bench.py --config $c --steps 2000 --reps 3 --traffic-summary $OUT/r05d_${n}_pmc_summary.json > $OUT/bench_${c}.json 2>>$OUT/bench_${c}.err
  show $OUT/bench_${c}.json "$c"
done
tail -n 3 $OUT/*.err
