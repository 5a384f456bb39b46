#!/bin/bash
# Round 5, sixth GPU call: BASELINE config 5 WHOLE on one GPU (17 x 262 144 in mode B), the six-building C3 line.
set -u
OUT=gpurun_out/r05f; mkdir -p $OUT; export TMPDIR=/tmp
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],'value %.3e'%d['value'],'launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],r['kernel'])" "$1" "$2"; }
python bench.py --config C5 --envs-per-gpu 262144 > $OUT/bench_C5_262144.json 2>$OUT/c5.err; show $OUT/bench_C5_262144.json "C5 whole"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5w -o run -- python bench.py --config C5 --envs-per-gpu 262144 --reps 1 > /dev/null 2>$OUT/trace_c5w.log
cp $OUT/trace_c5w/*kernel_stats.csv $OUT/c5_262144_kernel_stats.csv
python scripts/check_profiles.py --duration-tol 0.05 $OUT/bench_C5_262144.json $OUT/c5_262144_kernel_stats.csv
python bench.py --config C5 --envs-per-gpu 262144 --f64-chain > $OUT/bench_chain_C5_262144.json 2>/dev/null; show $OUT/bench_chain_C5_262144.json "C5 whole chain"
python bench.py --config C3-6 > $OUT/bench_C3-6.json 2>$OUT/c36.err; show $OUT/bench_C3-6.json "C3-6"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c36 -o run -- python bench.py --config C3-6 --reps 1 > /dev/null 2>$OUT/trace_c36.log
cp $OUT/trace_c36/*kernel_stats.csv $OUT/c36_kernel_stats.csv
python scripts/check_profiles.py $OUT/bench_C3-6.json $OUT/c36_kernel_stats.csv
tail -n 3 $OUT/*.err
