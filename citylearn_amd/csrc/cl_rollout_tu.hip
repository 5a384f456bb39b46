// cl_rollout_tu.hip -- the translation unit of the fused K-step rollout kernel (cl_rollout.h): cl_kernels.hip reduced to what that
// kernel needs plus its launcher `cl_tu_launch_rollout` (hidden visibility: not part of the C-ABI), compiled with
// -fno-slp-vectorize (citylearn_amd/_lib.py) -- see the comment at the launcher's declaration in cl_kernels.hip.
#define CL_TU_ROLLOUT
#include "cl_kernels.hip"
