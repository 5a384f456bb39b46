#!/bin/bash
# Round 5, seventh GPU call: the building-major streaming kernel (cl_tuning.envmajor = 3) against the env-major one.
set -u
OUT=gpurun_out/r05g; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "env_major" > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log); tail -5 $OUT/tests.log
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],'value %.3e'%d['value'],'launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],r['kernel'])" "$1" "$2"; }
S="--no-cpu-baseline --no-traffic-pass --no-streaming --no-chain-entry --steps 20 --warmup 5 --reps 3"
for i in 1 2 3; do
  python bench.py $S --envs-per-gpu 1048576 > $OUT/s_env_$i.json 2>/dev/null; show $OUT/s_env_$i.json "2^20 env-major"
  CL_TUNE_ENVMAJOR=3 python bench.py $S --envs-per-gpu 1048576 > $OUT/s_stream_$i.json 2>$OUT/stream.err; show $OUT/s_stream_$i.json "2^20 stream"
done
for E in 262144 524288 131072; do
  python bench.py $S --envs-per-gpu $E > $OUT/e_env_$E.json 2>/dev/null; show $OUT/e_env_$E.json "$E env-major"
  CL_TUNE_ENVMAJOR=3 python bench.py $S --envs-per-gpu $E > $OUT/e_stream_$E.json 2>/dev/null; show $OUT/e_stream_$E.json "$E stream"
done
CL_TUNE_ENVMAJOR=3 CL_TUNE_NT_STORES=2 python bench.py $S --envs-per-gpu 1048576 > $OUT/s_stream_plain.json 2>/dev/null; show $OUT/s_stream_plain.json "2^20 stream plain stores"
tail -n 3 $OUT/stream.err
