#!/bin/bash
# same-box A/B: the library as of the start of this session's last third (commit f5e7422) vs the final build, non-KPI step kernels
set -u
mkdir -p gpurun_out/r03_run31
for rep in 1 2; do
  for lib in "" citylearn_amd/libcl_alt_old.so; do
    CL_ALT_LIB=$lib timeout 200 python scripts/alt_lib_time.py lean thermal c3 c4 c4lean lean1m 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/r03_run31/final_vs_f5e7422_ab.log
