#!/bin/bash
# Round 5, third GPU call: the GPU suite with cl_dims.env_pitch, the HBM-streaming shape with and without the pitch.
set -u
OUT=gpurun_out/r05c; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 > $OUT/gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/gpu_tests.log)
tail -30 $OUT/gpu_tests.log
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],'value %.3e'%d['value'],'launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],r['kernel'],d['config'].get('env_pitch'))" "$1" "$2"; }
S="--no-cpu-baseline --no-traffic-pass --no-streaming --steps 20 --warmup 5 --reps 3"
for i in 1 2; do
  python bench.py $S --envs-per-gpu 1048576 > $OUT/s_pitch_$i.json 2>$OUT/s_pitch.err; show $OUT/s_pitch_$i.json "streaming 2^20, pitched"
  CL_BENCH_NO_PITCH=1 python bench.py $S --envs-per-gpu 1048576 > $OUT/s_nopitch_$i.json 2>/dev/null; show $OUT/s_nopitch_$i.json "streaming 2^20, plain"
done
python bench.py $S --envs-per-gpu 524288 > $OUT/s19_pitch.json 2>/dev/null; show $OUT/s19_pitch.json "2^19 pitched"
CL_BENCH_NO_PITCH=1 python bench.py $S --envs-per-gpu 524288 > $OUT/s19_nopitch.json 2>/dev/null; show $OUT/s19_nopitch.json "2^19 plain"
python bench.py $S --envs-per-gpu 262144 > $OUT/s18_pitch.json 2>/dev/null; show $OUT/s18_pitch.json "2^18 pitched"
CL_BENCH_NO_PITCH=1 python bench.py $S --envs-per-gpu 262144 > $OUT/s18_nopitch.json 2>/dev/null; show $OUT/s18_nopitch.json "2^18 plain"
python bench.py $S --envs-per-gpu 1048576 --f64-chain > $OUT/s_chain_pitch.json 2>/dev/null; show $OUT/s_chain_pitch.json "2^20 chain pitched"
python bench.py --no-cpu-baseline --no-traffic-pass --steps 2000 --reps 3 > $OUT/headline.json 2>$OUT/headline.err; show $OUT/headline.json "headline (+streaming inside)"
python -c "import json;d=json.load(open('$OUT/headline.json'));h=d['roofline']['hbm_streaming'];print('hbm_streaming', h['launch_us'], h['frac'], h['kernel'], h.get('env_pitch'))"
tail -n 3 $OUT/*.err
