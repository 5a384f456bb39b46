// cl_noslp_tu.hip -- the translation unit of the kernels that lose to SLP vectorisation (the fused K-step rollout kernel of cl_rollout.h
// and the plain lean step kernel): cl_kernels.hip reduced to what they need plus their launchers `cl_tu_launch_rollout` /
// `cl_tu_launch_lean` (hidden visibility: not part of the C-ABI), compiled with -fno-slp-vectorize (citylearn_amd/_lib.py) -- see the
// comment at the launchers' declaration in cl_kernels.hip.
#define CL_TU_NOSLP
#include "cl_kernels.hip"
