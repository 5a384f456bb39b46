#!/bin/bash
# Round 5, second GPU call: the GPU suite with CLD_F64_CHAIN, and what the chain costs per kernel.
set -u
OUT=gpurun_out/r05b; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 > $OUT/gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/gpu_tests.log)
tail -40 $OUT/gpu_tests.log
B="--no-cpu-baseline --no-traffic-pass --no-streaming --steps 2000 --reps 3"
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],'value %.3e'%d['value'],'launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],r['kernel'])" "$1" "$2"; }
python bench.py $B > $OUT/h_fp32.json 2>/dev/null; show $OUT/h_fp32.json "headline fp32"
python bench.py $B --f64-chain > $OUT/h_chain.json 2>$OUT/h_chain.err; show $OUT/h_chain.json "headline chain vec4"
CL_TUNE_VEC=2 python bench.py $B --f64-chain > $OUT/h_chain_v2.json 2>/dev/null; show $OUT/h_chain_v2.json "headline chain vec2"
python bench.py $B --f64-maps > $OUT/h_f64.json 2>/dev/null; show $OUT/h_f64.json "headline f64_maps"
python bench.py --no-cpu-baseline --no-traffic-pass --no-streaming --envs-per-gpu 1048576 --steps 20 --warmup 5 --reps 3 > $OUT/s_fp32.json 2>/dev/null; show $OUT/s_fp32.json "streaming fp32"
python bench.py --no-cpu-baseline --no-traffic-pass --no-streaming --envs-per-gpu 1048576 --steps 20 --warmup 5 --reps 3 --f64-chain > $OUT/s_chain.json 2>/dev/null; show $OUT/s_chain.json "streaming chain"
for c in T9 C4 C2; do
  python bench.py --config $c --steps 2000 --reps 3 > $OUT/${c}_fp32.json 2>/dev/null; show $OUT/${c}_fp32.json "$c fp32"
  python bench.py --config $c --steps 2000 --reps 3 --f64-chain > $OUT/${c}_chain.json 2>$OUT/${c}_chain.err; show $OUT/${c}_chain.json "$c chain"
done
python bench.py --config C5 > $OUT/C5_fp32.json 2>/dev/null; show $OUT/C5_fp32.json "C5 fp32"
python bench.py --config C5 --f64-chain > $OUT/C5_chain.json 2>$OUT/C5_chain.err; show $OUT/C5_chain.json "C5 chain"
python bench.py --config C3 > $OUT/C3_fp32.json 2>/dev/null; show $OUT/C3_fp32.json "C3 fp32"
python bench.py --config C3 --f64-chain > $OUT/C3_chain.json 2>$OUT/C3_chain.err; show $OUT/C3_chain.json "C3 chain"
tail -3 $OUT/*.err | head -40
