// Host build of citylearn_amd/csrc/cl_unit.h for CPU-side numerics debugging (TEST HARNESS ONLY, never a
// product path: the library itself has no CPU backend).
// g++ -O2 -shared -fPIC -DCL_HOST_SHIM -ffp-contract=off cl_unit_host.cpp -o libcl_unit_host.so
#define CL_HOST_SHIM 1
#include "../../citylearn_amd/csrc/cl_unit.h"

// state8: the six state planes of one unit + the low words of efficiency / degraded capacity (CLD_F64_MAPS)
template <bool FULL, int PREC>
static void run(const uint32_t* params, const float* ts_row, int t, int quirk, int rkind, const float* act6,
                float* state8, float* out10, float* reward) {
    cl::Bp B; cl::load_bp<FULL>(B, params);
    cl::Row R; cl::load_row<FULL>(R, ts_row, B.flags);
    cl::State S = {state8[0], state8[1], state8[2], state8[3], state8[4], state8[5], state8[6], state8[7]};
    cl::Act a = {act6[0], act6[1], act6[2], act6[3], act6[4], act6[5]};
    cl::Out O;
    cl::unit_step<FULL, PREC>(B, R, t, quirk != 0, a, S, O);
    *reward = cl::unit_reward<FULL>(rkind, B, S, O.net);
    state8[0] = S.soc; state8[1] = S.eff; state8[2] = S.degcap; state8[3] = S.cs; state8[4] = S.hs; state8[5] = S.ds;
    state8[6] = S.eff_lo; state8[7] = S.deg_lo;
    out10[0] = O.net; out10[1] = O.cost; out10[2] = O.emission; out10[3] = O.eb; out10[4] = O.cool_dem;
    out10[5] = O.c_cool; out10[6] = O.c_heat; out10[7] = O.c_dhw; out10[8] = O.c_ns; out10[9] = O.base_net;
}

extern "C" void host_unit_step(const uint32_t* params, const float* ts_row, int t, int quirk, int rkind, int full, int f64,
                               const float* act6, float* state8, float* out10, float* reward) {
    // f64: 0 = fp32 battery map, 1 = CLD_F64_MAPS, 2 = CLD_F64_CHAIN (state8[2] then holds the capacity loss)
    if (full && f64 == 2) run<true, 2>(params, ts_row, t, quirk, rkind, act6, state8, out10, reward);
    else if (full && f64) run<true, 1>(params, ts_row, t, quirk, rkind, act6, state8, out10, reward);
    else if (full) run<true, 0>(params, ts_row, t, quirk, rkind, act6, state8, out10, reward);
    else if (f64 == 2) run<false, 2>(params, ts_row, t, quirk, rkind, act6, state8, out10, reward);
    else if (f64) run<false, 1>(params, ts_row, t, quirk, rkind, act6, state8, out10, reward);
    else run<false, 0>(params, ts_row, t, quirk, rkind, act6, state8, out10, reward);
}

// CLD_CHECK: the same unit with the reference's runtime assertions compiled in; returns the CLV_* bits (include/citylearn_amd.h)
extern "C" unsigned host_unit_step_check(const uint32_t* params, const float* ts_row, int t, int quirk, int full, int f64, const float* act6, float* state8) {
    cl::State S = {state8[0], state8[1], state8[2], state8[3], state8[4], state8[5], state8[6], state8[7]};
    cl::Act a = {act6[0], act6[1], act6[2], act6[3], act6[4], act6[5]};
    cl::Out O;
    O.viol = 0u;
    cl::Bp B;
    cl::Row R;
    if (full) {
        cl::load_bp<true>(B, params); cl::load_row<true>(R, ts_row, B.flags);
        if (f64 == 2) cl::unit_step<true, 2, true>(B, R, t, quirk != 0, a, S, O);
        else if (f64) cl::unit_step<true, 1, true>(B, R, t, quirk != 0, a, S, O);
        else cl::unit_step<true, 0, true>(B, R, t, quirk != 0, a, S, O);
    } else {
        cl::load_bp<false>(B, params); cl::load_row<false>(R, ts_row, B.flags);
        if (f64 == 2) cl::unit_step<false, 2, true>(B, R, t, quirk != 0, a, S, O);
        else if (f64) cl::unit_step<false, 1, true>(B, R, t, quirk != 0, a, S, O);
        else cl::unit_step<false, 0, true>(B, R, t, quirk != 0, a, S, O);
    }
    state8[0] = S.soc; state8[1] = S.eff; state8[2] = S.degcap; state8[3] = S.cs; state8[4] = S.hs; state8[5] = S.ds;
    state8[6] = S.eff_lo; state8[7] = S.deg_lo;
    return O.viol;
}

// div_rn(a, b, RN(1 / b)) against the hardware division, element by element; returns the number of mismatching quotients
extern "C" long host_div_rn_mismatches(const double* a, const double* b, long n) {
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        const double q = cl::div_rn(a[i], b[i], 1.0 / b[i]), r = a[i] / b[i];
        bad += !(q == r || (q != q && r != r));
    }
    return bad;
}
