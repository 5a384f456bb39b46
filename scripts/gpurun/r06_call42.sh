#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== default build"; python scripts/r06_lean_streaming.py 2>&1 | grep -v amdgpu.ids
echo "== -DCL_EXP_NTL_ALL"; CITYLEARN_AMD_LIB=citylearn_amd/libcitylearn_amd_ntl.so python scripts/r06_lean_streaming.py 2>&1 | grep -v amdgpu.ids
