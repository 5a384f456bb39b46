#!/bin/bash
# Round 5, twenty-first GPU call: plain vs non-temporal loads x non-temporal stores (by footprint / never) on the latency-ordered lean kernel.
set -u
OUT=gpurun_out/r05v; mkdir -p $OUT; export TMPDIR=/tmp
ALT=$PWD/citylearn_amd/libcl_plainloads.so
run() { # E tag extra-args -- env...
  local E=$1 tag=$2 extra=$3; shift 3
  env "$@" python bench.py --envs-per-gpu $E --no-cpu-baseline --no-streaming --no-traffic-pass --no-chain-entry --steps 2000 --warmup 200 $extra > $OUT/h_${E}_$tag.json 2>$OUT/h_${E}_$tag.err || { echo "$E $tag FAILED"; return; }
  python -c "
import json
d=json.load(open('$OUT/h_${E}_$tag.json')); r=d['roofline']
print('headline', $E, '$tag', 'launch_us %.3f'%r['launch_us'], 'frac %.3f'%r['frac'], r['kernel'])
"
}
for rep in a b; do
  run 65536 ntld_ntst_$rep ""
  run 65536 plld_ntst_$rep "" CITYLEARN_AMD_LIB=$ALT
  run 65536 ntld_plst_$rep "" CL_TUNE_NT_STORES=2
  run 65536 plld_plst_$rep "" CITYLEARN_AMD_LIB=$ALT CL_TUNE_NT_STORES=2
done
for E in 8192 24576 32768 49152; do
  run $E ntld_a ""; run $E plld_a "" CITYLEARN_AMD_LIB=$ALT
  run $E ntld_b ""; run $E plld_b "" CITYLEARN_AMD_LIB=$ALT
done
run 65536 kpi_nt "--kpi"; run 65536 kpi_pl "--kpi" CITYLEARN_AMD_LIB=$ALT
run 65536 chain_nt "--f64-chain"; run 65536 chain_pl "--f64-chain" CITYLEARN_AMD_LIB=$ALT
python bench.py --config C2 --no-cpu-baseline > $OUT/c2_nt.json 2>/dev/null; CITYLEARN_AMD_LIB=$ALT python bench.py --config C2 --no-cpu-baseline > $OUT/c2_pl.json 2>/dev/null
python -c "
import json
for n in ('c2_nt','c2_pl'):
    d=json.load(open('$OUT/'+n+'.json')); print(n, d['roofline']['launch_us'])
"
