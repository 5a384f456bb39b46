// cl_lstm.h -- adjacent stage: LSTM indoor-temperature dynamics of `LSTMDynamicsBuilding`, one launch per env step.
// Included by cl_kernels.hip.
//
// Reference semantics (paths relative to /root/reference/citylearn/):
//   LSTMDynamicsBuilding._update_dynamics_input / get_dynamics_input / update_indoor_dry_bulb_temperature
//       building.py:3000-3078   (rolling window of lookback+1 = 13 normalised observation vectors; every feature uses
//       times t-11..t except the indoor temperature, which uses t-12..t-1; the prediction replaces the newest entry)
//   LSTMDynamics.forward        dynamics.py:95-101      (2-layer LSTM(13 -> 16), last time step, Linear(16 -> 1);
//       the hidden / cell state is carried from one env step's call to the next, building.py:3023-3024)
//
// Mapping: one wavefront = 64 envs of ONE building, so all 3.3 k weights of that building are wave-uniform: they are
// fetched with scalar loads and enter the v_fmac_f32 as the SGPR operand -- no LDS, no VGPRs for weights.  Eleven of
// the thirteen input features do not depend on the env (weather, calendar, set point, occupancy); their
// contribution to the layer-0 gates, W_ih0[:, exo] . x_exo(t) + b0, is precomputed on the host per (t, building)
// (`dyn_pre`), leaving 2 per-lane inputs (delivered cooling, previous indoor temperature) for layer 0.
// Per unit and env step: 12 window steps x (64 x 18 + 64 x 32) fused multiply-adds ~ 77 kFLOP, fp32 VALU-bound.
#pragma once

#define CL_LSTM_H 16            /* hidden size */
#define CL_LSTM_LOOKBACK 12
#define CL_LSTM_NW 3296         /* floats per building in `lstm_w` */
#define CL_LSTM_NPRE 80         /* floats per (t, building) in `dyn_pre` */
// lstm_w layout
#define CLW_WC 0                /* W_ih0[:, cooling_demand]  [64] */
#define CLW_WT 64               /* W_ih0[:, indoor_temperature] [64] */
#define CLW_WHH0 128            /* W_hh0 [64][16] */
#define CLW_WIH1 1152           /* W_ih1 [64][16] */
#define CLW_WHH1 2176           /* W_hh1 [64][16] */
#define CLW_B1 3200             /* b_ih1 + b_hh1 [64] */
#define CLW_WLIN 3264           /* Linear weight [16] */
#define CLW_BLIN 3280
#define CLW_TMIN 3281           /* indoor temperature normalisation */
#define CLW_TMAX 3282
#define CLW_CMIN 3283           /* cooling demand normalisation */
#define CLW_CMAX 3284
#define CLW_ACTIVE 3285         /* 1.0 if this building has a dynamics model */
#define CLW_RW_BAND 3286        /* ComfortReward band (NaN = use the data-file comfort band), exponents */
#define CLW_RW_LOEXP 3287
#define CLW_RW_HIEXP 3288
// dyn_pre layout: [0..63] layer-0 pre-gates, [64] data-file temperature (normalised), [65] data-file temperature [C]
#define CLPRE_TNORM 64
#define CLPRE_TRAW 65
#define CLPRE_HVAC 66           /* hvac_mode, cooling / heating set point, comfort band of the data file at t */
#define CLPRE_CSP 67
#define CLPRE_HSP 68
#define CLPRE_BAND 69

#ifdef __HIPCC__
namespace {

// Weights are read through the constant address space: loads from it are invariant by definition, so a wave-uniform
// address always becomes an s_load (through a plain global pointer the stores to `hist` / `hidden` make the compiler
// fall back to vector loads -- 3 k weights in VGPRs).
typedef const float __attribute__((address_space(4)))* cptr;
CL_DEV cptr as_const(const float* p) { return (cptr)(unsigned long long)p; }
// a zero the optimiser cannot see through: added to the weight base once per window step so that the (loop-invariant)
// weight loads are not hoisted out of the 12-step loop into thousands of live registers
CL_DEV int opaque_zero() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }

CL_DEV float sigmoidf_(float x) { return cl::rcp(1.0f + __expf(-x)); }
CL_DEV float tanhf_(float x) { return 2.0f * cl::rcp(1.0f + __expf(-2.0f * x)) - 1.0f; }

// One LSTM cell for 64 envs (one per lane).  gates[j] row order as in torch.nn.LSTM: i, f, g, o blocks of 16 rows.
// `pre(row)` = bias (+ env-independent input contribution); NX per-lane inputs x, weight of (row, k) at Wx[row*SR + k*SK].
template <int NX, int SR, int SK, typename Pre>
CL_DEV void lstm_cell(Pre pre, cptr Wx, const float (&x)[NX], cptr Wh,
                      float (&h)[CL_LSTM_H], float (&c)[CL_LSTM_H]) {
    float hn[CL_LSTM_H];
#pragma unroll
    for (int j = 0; j < CL_LSTM_H; ++j) {
        float g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = q * CL_LSTM_H + j;
            float acc = pre(row);
#pragma unroll
            for (int k = 0; k < NX; ++k) acc = fmaf(Wx[row * SR + k * SK], x[k], acc);
#pragma unroll
            for (int k = 0; k < CL_LSTM_H; ++k) acc = fmaf(Wh[row * CL_LSTM_H + k], h[k], acc);
            g[q] = acc;
        }
        const float cn = sigmoidf_(g[1]) * c[j] + sigmoidf_(g[0]) * tanhf_(g[2]);
        c[j] = cn;
        hn[j] = sigmoidf_(g[3]) * tanhf_(cn);
    }
#pragma unroll
    for (int j = 0; j < CL_LSTM_H; ++j) h[j] = hn[j];
}

// ComfortReward.calculate for one building (reward_function.py:269-334).
CL_DEV float comfort_reward(float temp, float cool_dem, float heat_dem, float mode, float csp, float hsp, float band,
                            float lo_exp, float hi_exp) {
    const bool heating = heat_dem > cool_dem;
    if (mode == 1.0f || mode == 2.0f) {
        const float sp = mode == 1.0f ? csp : hsp;
        const float delta = fabsf(temp - sp);
        if (temp < sp - band) return -__powf(delta, mode == 2.0f ? lo_exp : hi_exp);
        if (temp < sp) return heating ? 0.0f : -delta;
        if (temp <= sp + band) return heating ? -delta : 0.0f;
        return -__powf(delta, heating ? hi_exp : lo_exp);
    }
    const float cd = fabsf(temp - csp), hd = fabsf(temp - hsp);
    if (temp < hsp - band) return -__powf(hd, heating ? lo_exp : hi_exp);
    if (temp < hsp) return -hd;
    if (temp <= csp) return 0.0f;
    if (temp < csp + band) return -cd;
    return -__powf(cd, heating ? hi_exp : lo_exp);
}

struct LstmArgs {
    const float* __restrict__ lstm_w;     // [B][CL_LSTM_NW]
    const float* __restrict__ dyn_pre;    // [T][B][CL_LSTM_NPRE]
    const float* __restrict__ cool_dem;   // [B][E] delivered cooling of this step (out_bldg plane CLO_COOL_DEM)
    float* __restrict__ hist;             // [24][B][E]: rings of the last 12 normalised cooling demands / temperatures
    float* __restrict__ hidden;           // [64][B][E]: h0, c0, h1, c1
    float* __restrict__ indoor_temp;      // [B][E] out: indoor dry-bulb temperature of step t [C]
    const float* __restrict__ heat_dem;   // [B][E] delivered heating (may be NULL = 0)
    float* __restrict__ comfort;          // [B][E] out: ComfortReward of step t (may be NULL)
    int n_env, n_bldg, t;
};

CL_DEV float lstm_predict(const LstmArgs& a, cptr W, int b, long long plane, long long off) {
    float h0[CL_LSTM_H], c0[CL_LSTM_H], h1[CL_LSTM_H], c1[CL_LSTM_H];
#pragma unroll
    for (int j = 0; j < CL_LSTM_H; ++j) {
        h0[j] = a.hidden[(long long)(0 * CL_LSTM_H + j) * plane + off];
        c0[j] = a.hidden[(long long)(1 * CL_LSTM_H + j) * plane + off];
        h1[j] = a.hidden[(long long)(2 * CL_LSTM_H + j) * plane + off];
        c1[j] = a.hidden[(long long)(3 * CL_LSTM_H + j) * plane + off];
    }
    for (int s = 0; s < CL_LSTM_LOOKBACK; ++s) {
        const int time = a.t - (CL_LSTM_LOOKBACK - 1) + s;                    // every feature but the temperature: t-11 .. t
        const cptr pre = as_const(a.dyn_pre + ((long long)time * a.n_bldg + b) * CL_LSTM_NPRE);
        const cptr Ws = W + opaque_zero();
        float x[2];
        x[0] = a.hist[(long long)(time % CL_LSTM_LOOKBACK) * plane + off];                         // cooling demand at `time`
        x[1] = a.hist[(long long)(CL_LSTM_LOOKBACK + (time - 1) % CL_LSTM_LOOKBACK) * plane + off]; // temperature at `time - 1`
        lstm_cell<2, 1, 64>([&](int row) { return pre[row]; }, Ws + CLW_WC, x, Ws + CLW_WHH0, h0, c0);
        lstm_cell<CL_LSTM_H, CL_LSTM_H, 1>([&](int row) { return Ws[CLW_B1 + row]; }, Ws + CLW_WIH1, h0, Ws + CLW_WHH1, h1, c1);
    }
    float y = W[CLW_BLIN];
#pragma unroll
    for (int k = 0; k < CL_LSTM_H; ++k) y = fmaf(W[CLW_WLIN + k], h1[k], y);
#pragma unroll
    for (int j = 0; j < CL_LSTM_H; ++j) {
        a.hidden[(long long)(0 * CL_LSTM_H + j) * plane + off] = h0[j];
        a.hidden[(long long)(1 * CL_LSTM_H + j) * plane + off] = c0[j];
        a.hidden[(long long)(2 * CL_LSTM_H + j) * plane + off] = h1[j];
        a.hidden[(long long)(3 * CL_LSTM_H + j) * plane + off] = c1[j];
    }
    return y;
}

__global__ void __launch_bounds__(256) cl_lstm_kernel(const LstmArgs a) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int e = (blockIdx.x * 4 + wv) * 64 + lane;
    if (e >= a.n_env) return;
    const long long plane = (long long)a.n_bldg * a.n_env;
    const long long off = (long long)b * a.n_env + e;
    const cptr W = as_const(a.lstm_w + (long long)b * CL_LSTM_NW);
    const cptr pre_t = as_const(a.dyn_pre + ((long long)a.t * a.n_bldg + b) * CL_LSTM_NPRE);
    const float cool = a.cool_dem[off];
    float temp = pre_t[CLPRE_TRAW];
    if (W[CLW_ACTIVE] != 0.0f) {
        const float tmin = W[CLW_TMIN], tmax = W[CLW_TMAX], cmin = W[CLW_CMIN], cmax = W[CLW_CMAX];
        const int slot = a.t % CL_LSTM_LOOKBACK;
        // newest cooling-demand sample enters its ring (building.py:3068-3078)
        a.hist[(long long)slot * plane + off] = (cool - cmin) / (cmax - cmin);
        float y = pre_t[CLPRE_TNORM];      // warm-up: no prediction yet, the window keeps the data-file temperature
        if (a.t >= CL_LSTM_LOOKBACK) {     // lookback + 1 samples exist (building.py:2996-2999)
            y = lstm_predict(a, W, b, plane, off);
            temp = fmaf(y, tmax - tmin, tmin);                                          // building.py:3031-3037
        }
        a.hist[(long long)(CL_LSTM_LOOKBACK + slot) * plane + off] = y;                 // building.py:3027-3028
    }
    a.indoor_temp[off] = temp;
    if (a.comfort) {
        const float band_p = W[CLW_RW_BAND];
        const float band = band_p == band_p ? band_p : pre_t[CLPRE_BAND];
        a.comfort[off] = comfort_reward(temp, cool, a.heat_dem ? a.heat_dem[off] : 0.0f, pre_t[CLPRE_HVAC], pre_t[CLPRE_CSP],
                                        pre_t[CLPRE_HSP], band, W[CLW_RW_LOEXP], W[CLW_RW_HIEXP]);
    }
}

}  // namespace
#endif
