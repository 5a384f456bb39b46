#!/bin/bash
set -u
export TMPDIR=/tmp
for i in 1 2 3; do
  python scripts/r06_stream_ab.py 2>&1 | grep -v amdgpu.ids
  CITYLEARN_AMD_LIB=citylearn_amd/libcitylearn_amd_ntl.so python scripts/r06_stream_ab.py 2>&1 | grep -v amdgpu.ids
done
