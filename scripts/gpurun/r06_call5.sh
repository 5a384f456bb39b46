#!/bin/bash
# Round 6, fifth GPU call: the selection map on the final rules (-> tests/golden/kernel_selection_r06.json), then the round's profile recipe.
set -u
OUT=gpurun_out/r06e; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/cliffs_chain.jsonl
timeout 1500 python scripts/r06_cliffs.py $OUT/cliffs_chain.jsonl chain > $OUT/cliffs_chain.log 2>&1; echo "cliffs rc=$?"; grep -c . $OUT/cliffs_chain.jsonl
timeout 4200 bash scripts/profile_round.sh r06 > $OUT/profile_round.log 2>&1; echo "profile rc=$?"; tail -60 $OUT/profile_round.log
