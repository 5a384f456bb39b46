#!/bin/bash
set -u
OUT=gpurun_out/r03_run7
mkdir -p $OUT
timeout 1200 bash scripts/noslp_ab.sh 2 > $OUT/noslp_ab.log 2>&1
cat $OUT/noslp_ab.log
