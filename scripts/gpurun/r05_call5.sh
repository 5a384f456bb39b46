#!/bin/bash
# Round 5, fifth GPU call: chunk-major (swapped) grids against env-tile-major ones, alternating on one box (two builds of the library).
set -u
OUT=gpurun_out/r05e; mkdir -p $OUT; export TMPDIR=/tmp
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],'value %.3e'%d['value'],'launch_us %.2f'%r['launch_us'],'frac %.3f'%r['frac'],r['kernel'])" "$1" "$2"; }
for i in 1 2 3; do
  for c in C4 C4-lean; do
    python bench.py --config $c --steps 2000 --reps 3 > $OUT/${c}_swap_$i.json 2>/dev/null; show $OUT/${c}_swap_$i.json "$c chunk-major"
    CITYLEARN_AMD_LIB=citylearn_amd/libcl_alt_noswap.so python bench.py --config $c --steps 2000 --reps 3 > $OUT/${c}_noswap_$i.json 2>/dev/null; show $OUT/${c}_noswap_$i.json "$c tile-major"
  done
done
for c in C4 C4-lean; do
  python bench.py --config $c --envs-per-gpu 8192 --steps 1000 --reps 3 > $OUT/${c}_8192_swap.json 2>/dev/null; show $OUT/${c}_8192_swap.json "$c 8192 chunk-major"
  CITYLEARN_AMD_LIB=citylearn_amd/libcl_alt_noswap.so python bench.py --config $c --envs-per-gpu 8192 --steps 1000 --reps 3 > $OUT/${c}_8192_noswap.json 2>/dev/null; show $OUT/${c}_8192_noswap.json "$c 8192 tile-major"
done
