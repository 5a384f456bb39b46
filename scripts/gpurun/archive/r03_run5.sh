#!/bin/bash
set -u
OUT=gpurun_out/r03_run5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
P=gpurun_out/prof_r03; mkdir -p $P
python bench.py --config C4-lean > $P/bench_C4-lean.json 2>$P/bench_C4-lean.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_c4lean -o run -- python bench.py --config C4-lean --reps 1 > /dev/null 2>$P/trace_c4lean.log
cp $P/trace_c4lean/*kernel_stats.csv $P/c4lean_kernel_stats.csv
KC=$(python -c "import json; print(json.load(open('$P/bench_C4-lean.json'))['roofline']['kernel'].split('+')[0])")
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $P/pmc_c4lean_$ctr -o run -- python bench.py --config C4-lean --steps 300 --warmup 50 --reps 1 --no-graph > /dev/null 2>$P/pmc_c4lean_$ctr.log
done
python scripts/pmc_summary.py $P/c4lean_pmc_summary.json "$KC" $P/pmc_c4lean_FETCH_SIZE/*counter_collection.csv $P/pmc_c4lean_WRITE_SIZE/*counter_collection.csv
python scripts/check_profiles.py $P/bench_C4-lean.json $P/c4lean_pmc_summary.json $P/c4lean_kernel_stats.csv
timeout 300 python scripts/finish_stress.py > $OUT/finish_stress.log 2>&1; tail -4 $OUT/finish_stress.log
