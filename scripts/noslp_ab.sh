#!/bin/bash
# Same-box A/B of the library built with and without SLP vectorisation (packed fp32), alternating builds: which kernels belong in
# csrc/cl_noslp_tu.hip?  Build the alternative first (CPU container):
#   python -c "from citylearn_amd import _lib; _lib.build_variant('citylearn_amd/libcl_alt_noslp.so', ['-fno-slp-vectorize'])"
# then on the GPU box:  bash scripts/noslp_ab.sh [reps=2] > gpurun_out/noslp_ab.log
# (round 2: rollout and the plain lean step kernel win 9 % / 4 % without packing and live in the no-SLP unit; LSTM and the chunked
#  thermal launches lose 5 % / 2 %; env-major and C3 unchanged.  Not yet measured per kernel: the lean launches with a fused epilogue.)
# A second alternative worth the same A/B: no packed fp32 instructions at all, including the explicit two-envs-per-lane vector
# arithmetic of the thermal kernels (319 v_pk_*_f32 in cl_step_full_kernel<2, ...>), which -fno-slp-vectorize leaves alone:
#   _lib.build_variant('citylearn_amd/libcl_alt_nopk.so', ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops'])
#   A=citylearn_amd/libcl_alt_nopk.so bash scripts/noslp_ab.sh
A=${A:-citylearn_amd/libcl_alt_noslp.so}
REPS=${1:-2}
for rep in $(seq $REPS); do
  for lib in "" $A; do
    echo "== ${lib:-product}"
    CL_ALT_LIB=$lib timeout 120 python scripts/alt_lib_time.py lean thermal c3 c4 c4lean lean1m 2>&1 | grep -v amdgpu.ids
    CL_ALT_LIB=$lib timeout 60 python scripts/kpi_cost_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
    CL_ALT_LIB=$lib timeout 60 python scripts/step_observe_bench.py 2>&1 | grep -v amdgpu.ids | tail -6
    CL_ALT_LIB=$lib timeout 60 python scripts/ev_step_bench.py 65536 2>&1 | grep -v amdgpu.ids | tail -4
  done
done
