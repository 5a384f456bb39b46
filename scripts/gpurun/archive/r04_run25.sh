#!/bin/bash
# planes-first loads in the general kernel on battery + PV districts: parity + same-box A/B against the round's previous build
set -u
timeout 900 python -m pytest tests/test_gpu_config_sizes.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do
  for lib in citylearn_amd/libcl_alt_old.so ""; do
    CL_ALT_LIB=$lib CL_TUNING=finish=3 timeout 200 python scripts/alt_lib_time.py c4lean 2>&1 | grep -v amdgpu.ids
  done
done
for lib in citylearn_amd/libcl_alt_old.so ""; do CL_ALT_LIB=$lib timeout 200 python scripts/alt_lib_time.py c4lean 2>&1 | grep -v amdgpu.ids; CL_ALT_LIB=$lib CL_TUNING=lean_variant=1 timeout 200 python scripts/alt_lib_time.py lean 2>&1 | grep -v amdgpu.ids; done
