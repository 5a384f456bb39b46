// Host build of citylearn_amd/csrc/cl_unit.h for CPU-side numerics debugging (TEST HARNESS ONLY, never a
// product path: the library itself has no CPU backend).
// g++ -O2 -shared -fPIC -DCL_HOST_SHIM -ffp-contract=off cl_unit_host.cpp -o libcl_unit_host.so
#define CL_HOST_SHIM 1
#include "../../citylearn_amd/csrc/cl_unit.h"

template <bool FULL>
static void run(const uint32_t* params, const float* ts_row, int t, int quirk, int rkind, const float* act6,
                float* state6, float* out10, float* reward) {
    cl::Bp B; cl::load_bp<FULL>(B, params);
    cl::Row R; cl::load_row<FULL>(R, ts_row, B.flags);
    cl::State S = {state6[0], state6[1], state6[2], state6[3], state6[4], state6[5]};
    cl::Act a = {act6[0], act6[1], act6[2], act6[3], act6[4], act6[5]};
    cl::Out O;
    cl::unit_step<FULL>(B, R, t, quirk != 0, a, S, O);
    *reward = cl::unit_reward<FULL>(rkind, B, S, O.net);
    state6[0] = S.soc; state6[1] = S.eff; state6[2] = S.degcap; state6[3] = S.cs; state6[4] = S.hs; state6[5] = S.ds;
    out10[0] = O.net; out10[1] = O.cost; out10[2] = O.emission; out10[3] = O.eb; out10[4] = O.cool_dem;
    out10[5] = O.c_cool; out10[6] = O.c_heat; out10[7] = O.c_dhw; out10[8] = O.c_ns; out10[9] = O.base_net;
}

extern "C" void host_unit_step(const uint32_t* params, const float* ts_row, int t, int quirk, int rkind, int full,
                               const float* act6, float* state6, float* out10, float* reward) {
    if (full) run<true>(params, ts_row, t, quirk, rkind, act6, state6, out10, reward);
    else run<false>(params, ts_row, t, quirk, rkind, act6, state6, out10, reward);
}
