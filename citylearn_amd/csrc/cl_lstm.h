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
// Mapping: see cl_lstm_kernel below -- the 64 x K gate pre-activations of a cell are a small GEMM over the env batch and
// run on the matrix cores: the recurrent products W h with split 16-bit operands (two f16 or three bf16 terms, fp32-level
// results), the K = 2 remainder of layer 0 on v_mfma_f32_32x32x2_f32 (exact fp32).  Eleven of the thirteen input features do
// not depend on the env (weather, calendar, set point, occupancy); their contribution to the layer-0 gates,
// W_ih0[:, exo] . x_exo(t) + b0, is precomputed on the host per (t, building) (`dyn_pre`) and seeds the accumulators,
// leaving 2 per-env inputs (delivered cooling, previous indoor temperature) for layer 0.
// Per unit and env step: 12 window steps x (64 x 18 + 64 x 32) fused multiply-adds ~ 77 kFLOP and 12 x 2 x 16 x 10 = 3 840
// transcendentals.  What bounds the kernel is the second number: a v_exp_f32 / v_rcp_f32 holds the SIMD's vector ALU for
// about 12 cycles against 4 for a plain operation (scripts/lstm_timeline.py: a lone wave needs 0.68 us for the 80
// transcendental + 165 plain operations of one cell pair), and co-resident waves share that ALU serially -- the older wave
// wins every arbitration, the younger one advances exactly as much as the older one loses.  The matrix products now hide
// behind the cell updates; per window step a wave issues ~350 vector operations, 160 of them transcendental.
// History: two VALU variants were measured first on MI355X at 3 buildings x 65 536 envs: lane = env with the weights
// streamed through SGPRs (555 us per step, scalar-load latency bound) and lane = (env, hidden unit) with the weights
// resident in 200 VGPRs and ds_swizzle broadcasts (467 us, dependent-FMA latency bound at 1-2 waves per SIMD); the
// f32-MFMA form 220 us; split bf16 176 us; this file 113 us.
#pragma once

#define CL_LSTM_H 16            /* hidden size */
#define CL_LSTM_LOOKBACK 12
#define CL_LSTM_NHIDDEN_ 64      /* h0, c0, h1, c1 per (building, env): hidden[b][e][64] */
#define CL_LSTM_NW 3360         /* floats per building in `lstm_w` */
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
#define CLW_ACTIVE 3285         /* 0: no dynamics model; 1: LSTM(.. -> <= 16, 2 layers), the matrix-core kernel; 2 / 3: any other
                                   shape with 1 / 2 layers, cl_lstm_generic_kernel (tables `gen_w`, `gen_pre`, `gen_hidden`) */
#define CLW_RW_BAND 3286        /* ComfortReward band (NaN = use the data-file comfort band), exponents */
#define CLW_RW_LOEXP 3287
#define CLW_RW_HIEXP 3288
#define CLW_DEM_HEAT 3290        /* != 0: the model's demand input is heating_demand (delivered heating plane), not cooling_demand */
#define CLW_DEM2 3291            /* != 0: the model takes BOTH demands -- cooling first, delivered heating as a second env-dependent input */
#define CLW_C2MIN 3292           /* normalisation of that second input */
#define CLW_C2MAX 3293
#define CLW_W2 3296              /* W_ih0[:, second demand input] [64] (matrix-core kernel; zeros without one) */
#define CLW_KPI_BAND 3289        /* comfort band of the discomfort KPIs (evaluate()'s scalar, citylearn.py:1191) */
// dyn_pre layout: [0..63] layer-0 pre-gates, [64] data-file temperature (normalised), [65] data-file temperature [C]
#define CLPRE_TNORM 64
#define CLPRE_TRAW 65
#define CLPRE_HVAC 66           /* hvac_mode, cooling / heating set point, comfort band of the data file at t */
#define CLPRE_CSP 67
#define CLPRE_HSP 68
#define CLPRE_BAND 69
#define CLPRE_OCC 70            /* occupant_count and power-outage signal of the data file at t (comfort KPIs) */
#define CLPRE_OUTAGE 71

#ifdef __HIPCC__
namespace {

CL_DEV float sigmoidf_(float x) { return cl::rcp(1.0f + __expf(-x)); }
// exp(-2 x) = exp2(x * (-2 log2 e)): one multiply instead of two, and bit-identical to __expf(-2.0f * x) = exp2((-2 x) * log2 e)
// because the factor 2 is exact
CL_DEV float tanhf_(float x) { return 2.0f * cl::rcp(1.0f + __builtin_amdgcn_exp2f(x * -2.885390043258667f)) - 1.0f; }

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
    const uint16_t* __restrict__ lstm_wb; // [B][CL_LSTM_NWB] bf16 split-weight fragments (see CL_LSTM_NWB) or null
    const float* __restrict__ dyn_pre;    // [T][B][CL_LSTM_NPRE]
    const float* __restrict__ cool_dem;   // [B][E] delivered cooling of this step (out_bldg plane CLO_COOL_DEM)
    float* __restrict__ hist;             // [24][B][E]: rings of the last 12 normalised cooling demands / temperatures
    float* __restrict__ hidden;           // [B][E][64]: h0[16], c0[16], h1[16], c1[16] of every (building, env)
    float* __restrict__ indoor_temp;      // [B][E] out: indoor dry-bulb temperature of step t [C]
    const float* __restrict__ heat_dem;   // [B][E] delivered heating (may be NULL = 0)
    float* __restrict__ comfort;          // [B][E] out: ComfortReward of step t (may be NULL)
    float* __restrict__ kpi_comfort;      // [CL_NKC][B][E] streaming discomfort accumulators (may be NULL)
    const int32_t* __restrict__ env_row0;  // per-env-block episode offsets (cl_dims.env_row0) or null
    int n_env, n_bldg, t;
};

// What every variant of the stage leaves per (building, env): the indoor temperature, ComfortReward and the streaming
// discomfort KPIs.
// (`heat`: the delivered heating of this step, 0 without a heating plane -- loaded by the caller, early)
CL_DEV void lstm_outputs(const LstmArgs& a, const float* __restrict__ W, const float* __restrict__ pre_t, long long off, long long plane,
                         float temp, float cool, float heat) {
    // (the ten accumulators are fetched together, before the first store of this function: as `k[..] += ..` statements one after the
    //  other every load waits for the store in front of it -- ten dependent round trips at the end of every wave)
    float kc[CL_NKC];
    if (a.kpi_comfort) {
        const float* __restrict__ k = a.kpi_comfort + off;
#pragma unroll
        for (int p = 0; p < CL_NKC; ++p) kc[p] = k[p * plane];
    }
    a.indoor_temp[off] = temp;
    if (a.comfort) {
        const float band_p = W[CLW_RW_BAND];
        const float band = band_p == band_p ? band_p : pre_t[CLPRE_BAND];
        a.comfort[off] = comfort_reward(temp, cool, heat, pre_t[CLPRE_HVAC], pre_t[CLPRE_CSP],
                                        pre_t[CLPRE_HSP], band, W[CLW_RW_LOEXP], W[CLW_RW_HIEXP]);
    }
    if (a.kpi_comfort) {
        // CostFunction.discomfort / one_minus_thermal_resilience as running sums (cost_function.py:224-353):
        // deltas count only while the building is occupied; band = evaluate()'s scalar comfort band
        const bool occupied = pre_t[CLPRE_OCC] > 0.0f;
        const float band = W[CLW_KPI_BAND];
        const float cd = occupied ? temp - pre_t[CLPRE_CSP] : 0.0f, hd = occupied ? temp - pre_t[CLPRE_HSP] : 0.0f;
        const bool hot = cd > band, cold = hd < -band;
        const float cmag = fabsf(fminf(hd, 0.0f)), hmag = fabsf(fmaxf(cd, 0.0f));
        kc[CLKC_UNMET] += (hot || cold) ? 1.0f : 0.0f;
        kc[CLKC_COLD] += cold ? 1.0f : 0.0f;
        kc[CLKC_HOT] += hot ? 1.0f : 0.0f;
        kc[CLKC_COLD_MIN] = fminf(kc[CLKC_COLD_MIN], cmag);
        kc[CLKC_COLD_MAX] = fmaxf(kc[CLKC_COLD_MAX], cmag);
        kc[CLKC_COLD_SUM] += cmag;
        kc[CLKC_HOT_MIN] = fminf(kc[CLKC_HOT_MIN], hmag);
        kc[CLKC_HOT_MAX] = fmaxf(kc[CLKC_HOT_MAX], hmag);
        kc[CLKC_HOT_SUM] += hmag;
        kc[CLKC_UNMET_OUTAGE] += ((hot || cold) && pre_t[CLPRE_OUTAGE] != 0.0f) ? 1.0f : 0.0f;
        float* __restrict__ k = a.kpi_comfort + off;
#pragma unroll
        for (int p = 0; p < CL_NKC; ++p) k[p * plane] = kc[p];
    }
}

// ---- the gate pre-activations on the matrix cores ----------------------------------------------------------------
// G[64 gates x 32 envs] = W[64 x K] . X[K x 32 envs] per cell, two 32-row blocks {i, f} and {g, o}.  A wavefront owns 32
// envs of one building.  In the C/D layout (col = lane & 31 = env, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) a lane
// ends up with gates i, f, g, o of the SAME eight hidden units u(m) = (m & 3) + 8 (m >> 2) + 4 (lane >> 5), so the
// cell update is lane-local; and because the K order of a dot product is free, k-slot (2 kk + (lane >> 5)) is DEFINED
// to be hidden unit u(kk): the B operand of MFMA kk is then simply the lane's own h[kk] -- no cross-lane traffic at
// all.  The A operands (the weights, K-permuted the same way) are loaded once per wave (f16 split: 48 VGPRs).
typedef float f32x16 __attribute__((ext_vector_type(16)));

CL_DEV int lstm_unit(int m, int hh) { return (m & 3) + 8 * (m >> 2) + 4 * hh; }

template <int DBG = 0>
CL_DEV void lstm_act(const f32x16& d0, const f32x16& d1, float (&c)[8], float (&h)[8]) {
    if constexpr (DBG & 1) {                     // timing experiment: no activations (keeps the data dependence)
#pragma unroll
        for (int m = 0; m < 8; ++m) { c[m] = d0[8 + m] + d1[m]; h[m] = d1[8 + m] * 0.001f + d0[m] * 0.001f; }
        return;
    }
    if constexpr (DBG & 32) {
        // lstm_variant 32 (what LSTMStage selects where dynamics.cell_update_bounds proves the bound; 117.9 -> 108.3 us at 3 x 65 536,
        // profiles/r03a_lstm_check_variant32.log): 7 instead of 10 transcendentals per unit and cell through common denominators,
        //   c' = [c (1 + e_i)(1 + e_g) + (1 - e_g)(1 + e_f)] / [(1 + e_f)(1 + e_i)(1 + e_g)],   h = (1 - e_c) / [(1 + e_o)(1 + e_c)],
        // e_x = 2^z_x.  The products must stay finite: z_i + z_f + z_g < 126 and z_o < 62 are properties of the weights (a bound the
        // host can compute when it packs the tables: |b| + sum |W_ih| x_max + sum |W_hh|; every 2023 model passes, one baeda model
        // does not), the cell-state path is clamped here (tanh(c) is -1 to fp32 precision long before 2^64).
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float ei = __builtin_amdgcn_exp2f(d0[m]), ef = __builtin_amdgcn_exp2f(d0[8 + m]);
            const float eg = __builtin_amdgcn_exp2f(d1[m]), eo = __builtin_amdgcn_exp2f(d1[8 + m]);
            const float pf = 1.0f + ef, t = (1.0f + ei) * (1.0f + eg);
            const float cn = fmaf(c[m], t, (1.0f - eg) * pf) * cl::rcp(pf * t);
            c[m] = cn;
            const float ec = __builtin_amdgcn_exp2f(fminf(cn * -2.885390043258667f, 64.0f));
            h[m] = (1.0f - ec) * cl::rcp((1.0f + eo) * (1.0f + ec));
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        // the gate rows arrive pre-multiplied by -log2 e (i, f, o) / -2 log2 e (g) (dynamics.pack_lstm): 2^z = exp(-x) / exp(-2 x)
        const float gi = cl::rcp(1.0f + __builtin_amdgcn_exp2f(d0[m])), gf = cl::rcp(1.0f + __builtin_amdgcn_exp2f(d0[8 + m]));
        const float gg = 2.0f * cl::rcp(1.0f + __builtin_amdgcn_exp2f(d1[m])) - 1.0f, go = cl::rcp(1.0f + __builtin_amdgcn_exp2f(d1[8 + m]));
        const float cn = gf * c[m] + gi * gg;                                                 // f c + i g
        c[m] = cn;
        h[m] = go * tanhf_(cn);                                                               // o tanh(c)
    }
}


// ---- split 16-bit operands ---------------------------------------------------------------------------------------
// The f32-input MFMA runs at the f32 VECTOR rate and -- measured -- does not overlap with VALU work: with it the kernel
// costs MFMA time + activation time (153 us + 74 us at 3 x 65 536).  The 16-bit matrix cores are 16x faster and run
// beside the VALU, so the recurrent products W h go through v_mfma_f32_32x32x16_{bf16,f16} with both operands split into
// a few 16-bit terms whose partial products are accumulated in fp32 (LstmSplit below).  The weights are split on the host
// (dynamics.pack_lstm_split), h is split in registers after every cell.  The two env-dependent layer-0 inputs and the
// pre-gates keep the exact f32 MFMA (K = 2); the layer-1 bias is the C operand of the first product of its chain.
// (Tried: weight fragments in LDS with a 3-waves-per-SIMD register budget -- with the pipelined loop 33 spilled VGPRs,
// 191 us vs 176 us; with one accumulator pair live at a time 7 spills, 177 us vs 166 us on the same box: more resident
// waves do not help, the SIMD's vector ALU is the shared resource.  Tried: a start offset (s_sleep) for every other
// resident workgroup so that one wave's MFMA chains meet the other's activations: 165 us without, 166 - 218 us with offsets of
// 0.5 - 4 us.  Both measured on the three-term bf16 kernel of round 1.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
/* lstm_wb layout (stride CL_LSTM_NWB 16-bit words per building): fragment f = (2 * matrix{hh0, ih1, hh1} + row_block) * T + term,
   T = 3 bf16 terms or (CLD_LSTM_F16) 2 f16 terms, then [lane][8]: W[32 row_block + (lane & 31)][unit u(j, lane >> 5)], j = 0..7 */

// SPLIT = 1: three bf16 terms per operand, the six partial products with i + j <= 2 (dropped terms <= 2^-24 |W||h|).
// SPLIT = 2: two f16 terms per operand (x = x0 + x1 + r, |r| <= 2^-22 |x|, or 2^-25 absolute where x1 is subnormal), the three
//            partial products with i + j <= 1: half the matrix-pipe time and 20 fewer VALU operations per cell for an error of
//            <= 3 * 2^-22 |W||h| -- a few fp32 roundings of the gate sums, which the reference's own fp32 accumulation also has.
template <int SPLIT> struct LstmSplit { typedef bf16x8 v8; typedef __bf16 elem; static constexpr int T = 3; };
template <> struct LstmSplit<2> { typedef f16x8 v8; typedef _Float16 elem; static constexpr int T = 2; };

template <int N, int TERMS, typename V8, typename E>
CL_DEV void lstm_split(const float (&h)[8], V8 (&t)[N]) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = h[j];
#pragma unroll
    for (int k = 0; k < TERMS; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const E q = (E)r[j];                            // round to nearest even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
            t[k][j] = q;
            if (k + 1 < TERMS) r[j] -= (float)q;           // exact
        }
    }
}

CL_DEV f32x16 lstm_mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
CL_DEV f32x16 lstm_mfma16(const f16x8& a, const f16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// acc_r = c_r + sum_{i + j < T} A_r,i B_j for the two row blocks r, smallest terms first; the two accumulators alternate so that
// consecutive MFMAs are independent.  `c0` / `c1` may be the accumulators themselves (continue a sum) or other registers (the
// first product of a chain reads its bias from there: no instruction spent on initialising the accumulators).
// DBG & 8 (experiment): two bf16 terms per operand, the three partial products with i + j <= 1 (dropped terms
// <= 2^-16 |W||h|): 134 us instead of 160 us at 3 x 65 536, but |dT| grows from 5.7e-6 to 1.1e-4 C over the 24 cells and
// ComfortReward (cubic in the temperature error) misses the 1e-4 tolerance by 1.3x -- not used.
template <int DBG, int T, typename V8>
CL_DEV void lstm_mma(const V8 (&A0)[T], const V8 (&A1)[T], const V8 (&B)[T], const f32x16& c0, const f32x16& c1, f32x16& acc0, f32x16& acc1) {
    if constexpr (DBG & 2) {                                     // timing experiment: no MFMA
        acc0 = c0; acc1 = c1; acc0[0] += (float)B[0][0]; acc1[0] += (float)B[1][0] + (float)B[T - 1][0]; return;
    }
#define CL_MMA2(I, J, C0, C1) acc0 = lstm_mfma16(A0[I], B[J], C0); acc1 = lstm_mfma16(A1[I], B[J], C1);
    if constexpr (T == 2 || (DBG & 8)) { CL_MMA2(1, 0, c0, c1) CL_MMA2(0, 1, acc0, acc1) CL_MMA2(0, 0, acc0, acc1) }
    else { CL_MMA2(2, 0, c0, c1) CL_MMA2(1, 1, acc0, acc1) CL_MMA2(0, 2, acc0, acc1) CL_MMA2(1, 0, acc0, acc1) CL_MMA2(0, 1, acc0, acc1) CL_MMA2(0, 0, acc0, acc1) }
#undef CL_MMA2
}

// Wave-timeline stamps of the diagnostic build (-DCL_TRACE, scripts/lstm_timeline.py): REFCLK (100 MHz) values parked in the lanes of
// one VGPR (a compare and a select: no memory access, no exec change, no new basic block) and written out once at the end.  `v` is a value the
// stamp must follow; the v_mov makes the wave really wait for it (an MFMA result is not there when the MFMA has issued).
#ifdef CL_TRACE
#define CL_LT_PUT(val, k) lt_stamp = (lane == (k)) ? (unsigned)(val) : lt_stamp
#define CL_LT(k, v) do { unsigned long long t_; float q_ = (v); \
    asm("v_mov_b32 %1, %1\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(q_)); (v) = q_; \
    CL_LT_PUT((unsigned)t_, (k)); } while (0)
#else
#define CL_LT(k, v)
#endif

template <int DBG, int SPLIT, bool TWO>
__global__ void __launch_bounds__(256) cl_lstm_kernel(const LstmArgs a) {
    typedef typename LstmSplit<SPLIT>::v8 v8;
    typedef typename LstmSplit<SPLIT>::elem elem;
    constexpr int NT = LstmSplit<SPLIT>::T;
    const int lane = threadIdx.x & 63;
    const int col = lane & 31, hh = lane >> 5;
    const int wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int e = (blockIdx.x * 4 + wv) * 32 + col;
    const bool live = e < a.n_env;
#ifdef CL_TRACE
    unsigned lt_stamp = 0u;
    { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); CL_LT_PUT((unsigned)t_, 0); }
#endif
    const int ec = live ? e : a.n_env - 1;
    const long long plane = (long long)a.n_bldg * a.n_env;
    const long long off = (long long)b * a.n_env + ec;
    const float* __restrict__ W = a.lstm_w + (long long)b * CL_LSTM_NW;
    const int row0 = a.env_row0 ? a.env_row0[(blockIdx.x * 128) / CL_ROW0_BLOCK] : 0;      // workgroup = 128 envs: uniform
    const float* __restrict__ pre_t = a.dyn_pre + ((long long)(a.t + row0) * a.n_bldg + b) * CL_LSTM_NPRE;
    const float act = W[CLW_ACTIVE];
    if (act >= 2.0f) return;                                     // another LSTM shape: cl_lstm_generic_kernel owns this building
    const float tmin = W[CLW_TMIN], tmax = W[CLW_TMAX], cmin = W[CLW_CMIN], cmax = W[CLW_CMAX];
    const float dem2 = W[CLW_DEM2];                              // (read here, with the other parameters: at its last use the load would be an exposed latency per wave)
    // the demand the model was trained on: delivered cooling, or delivered heating for a heating-driven model
    const float* __restrict__ dem_src = (W[CLW_DEM_HEAT] != 0.0f && a.heat_dem) ? a.heat_dem : a.cool_dem;
    const int slot = a.t % CL_LSTM_LOOKBACK;
    if (act == 0.0f || a.t < CL_LSTM_LOOKBACK) {                 // block-uniform: no dynamics model, or the window is still filling
        const float cool = a.cool_dem[off];                      // (lookback + 1 samples must exist, building.py:2996-2999)
        if (act != 0.0f && live && hh == 0) {
            a.hist[(long long)slot * plane + off] = (dem_src[off] - cmin) / (cmax - cmin);               // building.py:3068-3078
            a.hist[(long long)(CL_LSTM_LOOKBACK + slot) * plane + off] = pre_t[CLPRE_TNORM];            // building.py:3027-3028
            if (dem2 != 0.0f && a.heat_dem)                                                             // a model that takes both demands
                a.hist[(long long)(2 * CL_LSTM_LOOKBACK + slot) * plane + off] = (a.heat_dem[off] - W[CLW_C2MIN]) / (W[CLW_C2MAX] - W[CLW_C2MIN]);
        }
        if (live && hh == 0) lstm_outputs(a, W, pre_t, off, plane, pre_t[CLPRE_TRAW], cool, a.heat_dem ? a.heat_dem[off] : 0.0f);   // the data-file temperature
        return;
    }
    // From here on: one straight line up to the window loop.  Everything the wave needs from memory is requested first, in one
    // batch (weights, carried state, biases, this step's demand, the inputs of window steps 0 and 1), and only then used; left
    // where their values are first needed, the compiler requests the recurrent weights of layer 1 only after the first layer-0
    // products have waited for theirs.  (Measured: no gain by itself, 121 vs 119 us -- a wave's slow first window steps, 7 us
    // instead of 1.9 us, are the arbitration against the older wave of its SIMD, not its loads: scripts/lstm_timeline.py.)
    const float cool = a.cool_dem[off];
    const float dem = dem_src[off];
    const float heat = a.heat_dem ? a.heat_dem[off] : 0.0f;      // (ComfortReward, at the very end)
    float wlin[8];                                               // Linear(16 -> 1) weights of this lane's eight units
#pragma unroll
    for (int m = 0; m < 8; ++m) wlin[m] = W[CLW_WLIN + lstm_unit(m, hh)];
    const float blin = W[CLW_BLIN];
    float temp, y;
    // A operands: row = 32 rb + col of the torch gate matrix, k-slot 2 kk + hh -> hidden unit u(kk)
    float a_hh0[2][8], a_x0[2], a_x2[2], a_ih1[2][8], a_hh1[2][8];
    v8 A_hh0[2][NT], A_ih1[2][NT], A_hh1[2][NT];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int row = 32 * rb + col;
        a_x0[rb] = hh ? W[CLW_WT + row] : W[CLW_WC + row];
        a_x2[rb] = W[CLW_W2 + row];                          // (zeros unless the model takes a second demand input)
        if constexpr (SPLIT) {
            const v8* __restrict__ F = reinterpret_cast<const v8*>(a.lstm_wb + (long long)b * CL_LSTM_NWB);
#pragma unroll
            for (int k = 0; k < ((DBG & 8) ? 2 : NT); ++k) {
                A_hh0[rb][k] = F[((0 * 2 + rb) * NT + k) * 64 + lane];
                A_ih1[rb][k] = F[((1 * 2 + rb) * NT + k) * 64 + lane];
                A_hh1[rb][k] = F[((2 * 2 + rb) * NT + k) * 64 + lane];
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int u = lstm_unit(kk, hh);
                a_hh0[rb][kk] = W[CLW_WHH0 + row * CL_LSTM_H + u];
                a_ih1[rb][kk] = W[CLW_WIH1 + row * CL_LSTM_H + u];
                a_hh1[rb][kk] = W[CLW_WHH1 + row * CL_LSTM_H + u];
            }
        }
    }
    // carried state of this lane's eight units
    float* hid = a.hidden + ((long long)b * a.n_env + ec) * CL_LSTM_NHIDDEN_;
    float h0[8], c0[8], h1[8], c1[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int u = lstm_unit(m, hh);
        h0[m] = hid[0 * CL_LSTM_H + u]; c0[m] = hid[1 * CL_LSTM_H + u];
        h1[m] = hid[2 * CL_LSTM_H + u]; c1[m] = hid[3 * CL_LSTM_H + u];
    }
    // The window loop issues no memory instruction on its critical path: the env-independent layer-0 pre-gates enter the
    // accumulators through one extra MFMA (A = the 64 values, B = 1 in k-slot 0) instead of 32 loads + accumulator
    // writes per cell, and the three per-lane inputs of a step (two pre-gate values, one history sample) are fetched two
    // steps ahead of their use.
#define CL_MFMA(A, B, C) ((DBG & 2) ? (C) + (A) * (B) : __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, C, 0, 0, 0))
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // A model that takes BOTH demands (round 3): the pre-gate product below is a K = 2 MFMA whose second k-slot was idle (B = 0 on the
    // hh = 1 lanes).  The second demand input rides there -- A = its weight column, B = its normalised value at the window step -- so a
    // third env-dependent input costs no matrix instruction: one more ring read per lane and step, two selects -- which still cost 8 us
    // of 96 at 3 x 65 536 (the loads sit in the software pipeline's fetch slots), so the path is an instantiation of its own (TWO),
    // launched for districts whose caller sets CLD_LSTM_TWO_DEMANDS; in the other instantiation a building with such a model gets NaN.
    const float one_b = hh ? 0.0f : 1.0f;
    const bool two = TWO && dem2 != 0.0f && a.heat_dem;
    const float heat_n = two ? (heat - W[CLW_C2MIN]) / (W[CLW_C2MAX] - W[CLW_C2MIN]) : 0.0f;
    const float* __restrict__ hist2_lane = a.hist + off + 2ll * CL_LSTM_LOOKBACK * plane;
    const float a_b1[2] = {W[CLW_B1 + col], W[CLW_B1 + 32 + col]};
    // ring rows without a division or a branch in the loop: time % 12 = (m + s) mod 12, (time - 1) % 12 = that - 1 mod 12
    const int ring_m = (a.t - (CL_LSTM_LOOKBACK - 1)) % CL_LSTM_LOOKBACK;           // a.t >= 12 here
    const float* __restrict__ hist_lane = a.hist + off + (hh ? (long long)CL_LSTM_LOOKBACK * plane : 0ll);
    // The env-independent layer-0 pre-gates of a step are two values per lane and one extra K = 2 MFMA per row block.
    // (Tried: loading them straight in the C/D layout -- 16 values per row block and lane, four 16-byte loads each -- as the C
    // operand of the chain's first MFMA, the way the layer-1 bias enters: two MFMAs fewer per step but eight loads and 32
    // registers more: 119.2 vs 117.5 us with the f16 split, 168 vs 142 us with bf16 at 256 registers.)
    auto fetch = [&](int s, float (&ap)[2], float& xin, float& xb) {
        const int time = a.t - (CL_LSTM_LOOKBACK - 1) + s;
        const float* __restrict__ pre = a.dyn_pre + ((long long)(time + row0) * a.n_bldg + b) * CL_LSTM_NPRE;
        const float p0 = pre[col], p1 = pre[32 + col];
        // extra k-pair of layer 0: slot 0 = cooling demand at `time`, slot 1 = temperature at `time - 1`
        int r0 = ring_m + s; r0 -= r0 >= CL_LSTM_LOOKBACK ? CL_LSTM_LOOKBACK : 0;
        const int r1 = r0 == 0 ? CL_LSTM_LOOKBACK - 1 : r0 - 1;
        xin = hist_lane[(long long)(hh ? r1 : r0) * plane];        // (step 11, slot 0 is overridden at the point of use)
        if constexpr (TWO) {
            const float x2 = hist2_lane[(long long)r0 * plane];    // second demand input at `time` (step 11: produced by this launch)
            // pre-gate product: k-slot 0 = (pre-gates, 1), k-slot 1 = (weights of the second demand input, its value)
            ap[0] = hh ? a_x2[0] : p0; ap[1] = hh ? a_x2[1] : p1;
            xb = hh ? ((two && s == CL_LSTM_LOOKBACK - 1) ? heat_n : x2) : 1.0f;
        } else { ap[0] = p0; ap[1] = p1; xb = one_b; }
    };
    v8 H0[NT], H1[NT];                                            // split hidden states (B operands)
    auto split = [&](const float (&h)[8], v8 (&t)[NT]) { lstm_split<NT, (DBG & 8) ? 2 : NT, v8, elem>(h, t); };
    // layer-1 bias in the C/D layout of the two row blocks: the first product of every layer-1 chain reads it as its C operand
    f32x16 bias1[2];
    if constexpr (SPLIT) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) bias1[rb][r] = W[CLW_B1 + 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * hh];
    }
    auto layer0 = [&](const float (&ap)[2], float xin, float xb, f32x16& d0, f32x16& d1) {
        d0 = CL_MFMA(ap[0], TWO ? xb : one_b, zero16);
        d1 = CL_MFMA(ap[1], TWO ? xb : one_b, zero16);
        d0 = CL_MFMA(a_x0[0], xin, d0);
        d1 = CL_MFMA(a_x0[1], xin, d1);
        if constexpr (SPLIT) lstm_mma<DBG>(A_hh0[0], A_hh0[1], H0, d0, d1, d0, d1);
        else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                d0 = CL_MFMA(a_hh0[0][kk], h0[kk], d0);
                d1 = CL_MFMA(a_hh0[1][kk], h0[kk], d1);
            }
        }
    };
    f32x16 d0, d1, e0, e1;
    float ap[2], xin, xb, ap_n[2] = {0.0f, 0.0f}, xin_n = 0.0f, xb_n = 0.0f, ap_nn[2] = {0.0f, 0.0f}, xin_nn = 0.0f, xb_nn = 0.0f;
    fetch(0, ap, xin, xb);
    fetch(1, ap_n, xin_n, xb_n);
    __builtin_amdgcn_sched_barrier(0);                             // every load above is issued before anything below
    const float cool_n = (dem - cmin) / (cmax - cmin);
    if (live && hh == 0) a.hist[(long long)slot * plane + off] = cool_n;         // building.py:3068-3078
    if (live && hh == 0 && two) a.hist[(long long)(2 * CL_LSTM_LOOKBACK + slot) * plane + off] = heat_n;
    if constexpr (SPLIT) { split(h0, H0); split(h1, H1); }
    layer0(ap, xin, xb, d0, d1);
    CL_LT(1, d0[15]);                                             // weights, carried state and the first inputs arrived; first layer-0 gates done
    // The window loop, software-pipelined across the two layers.  Each layer is a strict chain matrix product -> cell update ->
    // matrix product, so inside one wave matrix-core work can only run beside the OTHER layer's cell update (and the second
    // wave of the SIMD does not fill the gaps: the older wave wins every arbitration, the younger one advances at a fifth of
    // the speed until the older one ends -- scripts/lstm_timeline.py).  Layer 1 therefore runs one step behind layer 0:
    //   phase A(s): cell update of layer 0, step s          beside   e(s-1) += W_hh1 h1(s-2)
    //   phase B(s): cell update of layer 1, step s-1        beside   e(s) = b1 + W_ih1 h0(s),  d(s+1) = layer-0 gates of step s+1
    // and every matrix product is issued next to a cell update that does not depend on it.  The window is 12 steps, so the
    // first and the last step are written out (no branch inside the loop: with an `if` around a stage the compiler sinks that
    // stage's loads into the conditional block, right in front of their use).
    auto fetch_ahead = [&](int s) {
        // the three per-lane inputs of layer 0 are fetched two steps before their use (rotated at the end of the step)
        fetch(min(s + 2, CL_LSTM_LOOKBACK - 1), ap_nn, xin_nn, xb_nn);
        __builtin_amdgcn_sched_barrier(0);                          // the loads stay first
    };
    auto rotate = [&](int s) {                                      // at the very end: the copies wait for this step's loads
        __builtin_amdgcn_sched_barrier(0);
        ap_n[0] = ap_nn[0]; ap_n[1] = ap_nn[1]; xin_n = xin_nn; xb_n = xb_nn;
        CL_LT(50 + s, xin_n);                                       // the inputs fetched at the top of this step have arrived
    };
    auto hh1_product = [&]() {                                      // e += W_hh1 h1
        if constexpr (SPLIT) lstm_mma<DBG>(A_hh1[0], A_hh1[1], H1, e0, e1, e0, e1);
        else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                e0 = CL_MFMA(a_hh1[0][kk], h1[kk], e0);
                e1 = CL_MFMA(a_hh1[1][kk], h1[kk], e1);
            }
        }
    };
    auto phase_a = [&](int s, auto has_h) {
        if constexpr (decltype(has_h)::value) hh1_product();
        lstm_act<DBG>(d0, d1, c0, h0);
        if constexpr (SPLIT) split(h0, H0);
        CL_LT(3 + 4 * s, h0[7]);                                    // layer-0 cell update of step s done
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase_b = [&](int s, auto has_e, auto has_l0) {
        const f32x16 g0 = e0, g1 = e1;                              // the complete layer-1 gates of step s - 1
        if constexpr (SPLIT) lstm_mma<DBG>(A_ih1[0], A_ih1[1], H0, bias1[0], bias1[1], e0, e1);
        else {
            e0 = CL_MFMA(a_b1[0], one_b, zero16);
            e1 = CL_MFMA(a_b1[1], one_b, zero16);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                e0 = CL_MFMA(a_ih1[0][kk], h0[kk], e0);
                e1 = CL_MFMA(a_ih1[1][kk], h0[kk], e1);
            }
        }
        // the newest cooling sample (step 11, k-slot 0) was produced by this launch, not read from the ring
        if constexpr (decltype(has_l0)::value) layer0(ap_n, (!hh && s + 1 == CL_LSTM_LOOKBACK - 1) ? cool_n : xin_n, xb_n, d0, d1);
        if constexpr (decltype(has_e)::value) {
            lstm_act<DBG>(g0, g1, c1, h1);
            if constexpr (SPLIT) split(h1, H1);
            CL_LT(5 + 4 * (s - 1), h1[7]);                          // layer-1 cell update of step s - 1 done
        }
    };
    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    fetch_ahead(0); phase_a(0, no); phase_b(0, no, yes); rotate(0);
    for (int s = 1; s < CL_LSTM_LOOKBACK - 1; ++s) { fetch_ahead(s); phase_a(s, yes); phase_b(s, yes, yes); rotate(s); }
    phase_a(CL_LSTM_LOOKBACK - 1, yes); phase_b(CL_LSTM_LOOKBACK - 1, yes, no);
    hh1_product();                                                  // drain: layer 1 of the last step
    lstm_act<DBG>(e0, e1, c1, h1);
    CL_LT(5 + 4 * (CL_LSTM_LOOKBACK - 1), h1[7]);
#undef CL_MFMA
    if (live) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int u = lstm_unit(m, hh);
            hid[0 * CL_LSTM_H + u] = h0[m]; hid[1 * CL_LSTM_H + u] = c0[m];
            hid[2 * CL_LSTM_H + u] = h1[m]; hid[3 * CL_LSTM_H + u] = c1[m];
        }
    }
    // Linear(16 -> 1): this lane's eight units, then the other half of the env (lane ^ 32)
    float part = 0.0f;
#pragma unroll
    for (int m = 0; m < 8; ++m) part = fmaf(wlin[m], h1[m], part);
    const float other = __shfl_xor(part, 32);
    y = blin + (hh ? other + part : part + other);
    temp = fmaf(y, tmax - tmin, tmin);                           // building.py:3031-3037
    if constexpr (!TWO) { if (dem2 != 0.0f) temp = __builtin_nanf(""); }     // a both-demand model in a launch without CLD_LSTM_TWO_DEMANDS: loud, not wrong
    if (live && hh == 0) a.hist[(long long)(CL_LSTM_LOOKBACK + slot) * plane + off] = y;   // building.py:3027-3028
    if (live && hh == 0) lstm_outputs(a, W, pre_t, off, plane, temp, cool, heat);
#ifdef CL_TRACE
    {
        unsigned long long t_;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
        CL_LT_PUT((unsigned)t_, 62);
        CL_LT_PUT(__builtin_amdgcn_s_getreg(63492) | (__builtin_amdgcn_s_getreg(63508) << 16), 63);   // HW_ID | XCC_ID << 16
        if (g_cl_trace) reinterpret_cast<unsigned*>(g_cl_trace)[(((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 64 + lane] = lt_stamp;
    }
#endif
}

// ---- any other LSTM shape (hidden size <= 64, one or two layers; or a model that takes both demands): plain fp32 FMAs ----------------
// baeda_3dem's Building_4 is LSTM(11 -> 50, one layer).  A workgroup = 64 envs of one building x CL_GEN_NWV waves; lane = env, and the
// hidden UNITS are dealt to the waves (wave v owns units v * UPW .. : UPW = ceil(H / CL_GEN_NWV) <= 16).  A wave keeps the four gate
// accumulators and the cell state of its units in registers, walks the input index k in the OUTER loop -- one LDS read of h[k] feeds
// 4 * UPW FMAs whose weights arrive by scalar loads (wave-uniform addresses) -- and only the hidden state travels through LDS
// ([2][H][64] per layer, double-buffered: one barrier per cell).
//   gen_w   [B][GW]       per building: WX [H][12] (gates i, f, g, o of the demand input, of the temperature input, of the second demand input),
//                         WHH0 [H][H][4], WIH1 [H][H][4], WHH1 [H][H][4], B1 [H][4], WLIN [H]   (H = the padded hidden size)
//   gen_pre [T][B][H][4]  env-independent part of the layer-0 gates of (t, building)
//   gen_hidden [B][4][H][E]  h0, c0, h1, c1 carried across env steps
// Padded units have zero weights: their state stays 0.
// Round 3: the first version ran ONE wave per 64 envs with every state in LDS and the unit index in the outer loop (H^2 LDS reads per
// cell, one wave per SIMD at 65 536 envs: nothing to hide its LDS and scalar-load latencies behind): 4.5 ms per step for baeda_3dem at
// 4 x 65 536, 0.78 ms for a 2 x 16-unit model (profiles/r03c_lstm_generic_bench.log).
struct LstmGenArgs {
    LstmArgs s;
    const float* __restrict__ gen_w;
    const float* __restrict__ gen_pre;
    float* __restrict__ gen_hidden;
    int H;          // padded hidden size of the tables
    long long gw;   // floats per building in gen_w
};

constexpr int CL_GEN_NWV = 4;        // waves per workgroup
constexpr int CL_GEN_UPW = (CL_LSTM_GEN_HMAX + CL_GEN_NWV - 1) / CL_GEN_NWV;      // units per wave at most (16)

// gates[j][0..3] += W[u0 + j][k][0..3] * x[k] for k < H: x from LDS ([k][64] floats); W ([unit][input][gate]) from the workgroup's
// staged copy in LDS (STAGED: one broadcast ds_read_b128 per (unit, k) -- every lane reads the same address) or by scalar loads
template <bool STAGED>
CL_DEV void lstm_gen_matvec(float (&g)[CL_GEN_UPW][4], const float* wmat, const float* xs, int H, int u0, int n_u, int lane) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int k = 0; k < H; ++k) {
        const float x = xs[k * 64 + lane];
#pragma unroll
        for (int j = 0; j < CL_GEN_UPW; ++j) {
            if (j < n_u) {                                          // wave-uniform
                const float* q = wmat + ((long long)(u0 + j) * H + k) * 4;
                if constexpr (STAGED) {
                    const f4 w = *reinterpret_cast<const f4*>(q);
                    g[j][0] = fmaf(w[0], x, g[j][0]); g[j][1] = fmaf(w[1], x, g[j][1]);
                    g[j][2] = fmaf(w[2], x, g[j][2]); g[j][3] = fmaf(w[3], x, g[j][3]);
                } else {
                    g[j][0] = fmaf(q[0], x, g[j][0]); g[j][1] = fmaf(q[1], x, g[j][1]);
                    g[j][2] = fmaf(q[2], x, g[j][2]); g[j][3] = fmaf(q[3], x, g[j][3]);
                }
            }
        }
    }
}

// the cell update of this wave's units: c in registers, new h into LDS
CL_DEV void lstm_gen_update(const float (&g)[CL_GEN_UPW][4], float (&c)[CL_GEN_UPW], float* h_new, int u0, int n_u, int lane) {
#pragma unroll
    for (int j = 0; j < CL_GEN_UPW; ++j) {
        if (j < n_u) {
            const float cn = sigmoidf_(g[j][1]) * c[j] + sigmoidf_(g[j][0]) * tanhf_(g[j][2]);      // f c + i g
            c[j] = cn;
            h_new[(u0 + j) * 64 + lane] = sigmoidf_(g[j][3]) * tanhf_(cn);                          // o tanh(c)
        }
    }
}

// STAGED: the recurrent matrices of the building are copied into LDS once per workgroup (they are re-read by all 12 window steps): a
// weight then costs a broadcast LDS read instead of a scalar load whose latency nothing hides -- with the weights through scalar loads a
// 2 x 16-unit model took 750 us per step at 3 x 65 536 (16 dependent scalar round trips per matrix-vector product).  The host stages
// whenever hidden state + matrices fit 150 KB of LDS (everything up to two layers of 45 units or one of 64).
template <bool STAGED>
__global__ void __launch_bounds__(64 * CL_GEN_NWV) cl_lstm_generic_kernel(const LstmGenArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // h0 [2][H][64], h1 [2][H][64] (two layers), then (STAGED) the matrices
    const LstmArgs& a = g.s;
    const int lane = threadIdx.x & 63, b = blockIdx.y, H = g.H;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int e = blockIdx.x * 64 + lane;
    const bool live = e < a.n_env;
    const int ec = live ? e : a.n_env - 1;
    const long long plane = (long long)a.n_bldg * a.n_env;
    const long long off = (long long)b * a.n_env + ec;
    const float* __restrict__ W = a.lstm_w + (long long)b * CL_LSTM_NW;
    const float mode = W[CLW_ACTIVE];
    if (mode < 2.0f) return;                                      // the matrix-core kernel (or nobody) owns this building (whole workgroup)
    const int layers = mode >= 3.0f ? 2 : 1;
    const int upw = (H + CL_GEN_NWV - 1) / CL_GEN_NWV;
    const int u0 = wv * upw, n_u = max(0, min(upw, H - u0));      // this wave's units
    const bool lead = wv == 0;                                    // the wave that owns the per-env side effects
    const int row0 = a.env_row0 ? a.env_row0[(blockIdx.x * 64) / CL_ROW0_BLOCK] : 0;
    const float* __restrict__ pre_t = a.dyn_pre + ((long long)(a.t + row0) * a.n_bldg + b) * CL_LSTM_NPRE;
    const float cool = a.cool_dem[off];
    float temp = pre_t[CLPRE_TRAW];
    const float tmin = W[CLW_TMIN], tmax = W[CLW_TMAX], cmin = W[CLW_CMIN], cmax = W[CLW_CMAX];
    const float dem = (W[CLW_DEM_HEAT] != 0.0f && a.heat_dem) ? a.heat_dem[off] : cool;
    const float cool_n = (dem - cmin) / (cmax - cmin);
    const int slot = a.t % CL_LSTM_LOOKBACK;
    if (live && lead) a.hist[(long long)slot * plane + off] = cool_n;                     // building.py:3068-3078
    // a model that takes both demands: delivered heating is its second env-dependent input, with a ring of its own (rows 24 .. 35)
    const bool two = W[CLW_DEM2] != 0.0f && a.heat_dem;
    const float heat_n = two ? (a.heat_dem[off] - W[CLW_C2MIN]) / (W[CLW_C2MAX] - W[CLW_C2MIN]) : 0.0f;
    if (live && lead && two) a.hist[(long long)(2 * CL_LSTM_LOOKBACK + slot) * plane + off] = heat_n;
    float y = pre_t[CLPRE_TNORM];
    if (a.t >= CL_LSTM_LOOKBACK) {
        float* h0b[2] = {lds, lds + H * 64};
        float* h1b[2] = {lds + 2 * H * 64, lds + 3 * H * 64};              // (one layer: aliases the staging area, never touched)
        float* hid = g.gen_hidden + ((long long)b * 4 * H) * a.n_env + ec;
        float c0[CL_GEN_UPW], c1[CL_GEN_UPW];
#pragma unroll
        for (int j = 0; j < CL_GEN_UPW; ++j) {
            c0[j] = c1[j] = 0.0f;
            if (j < n_u) {
                const int u = u0 + j;
                h0b[0][u * 64 + lane] = hid[(long long)(0 * H + u) * a.n_env]; c0[j] = hid[(long long)(1 * H + u) * a.n_env];
                if (layers == 2) { h1b[0][u * 64 + lane] = hid[(long long)(2 * H + u) * a.n_env]; c1[j] = hid[(long long)(3 * H + u) * a.n_env]; }
            }
        }
        const float* __restrict__ G = g.gen_w + (long long)b * g.gw;
        const float* wx = G, * whh0 = wx + H * 12, * wih1 = whh0 + (long long)H * H * 4, * whh1 = wih1 + (long long)H * H * 4;
        const float* b1 = whh1 + (long long)H * H * 4, * wlin = b1 + H * 4;
        const float* m_hh0 = whh0, * m_ih1 = wih1, * m_hh1 = whh1;
        if constexpr (STAGED) {
            float* stage = lds + (layers == 2 ? 4 : 2) * H * 64;
            const int n_w = (layers == 2 ? 3 : 1) * H * H * 4;                  // WHH0 (, WIH1, WHH1): contiguous in gen_w
            for (int i = threadIdx.x; i < n_w; i += blockDim.x) stage[i] = whh0[i];
            m_hh0 = stage; m_ih1 = stage + H * H * 4; m_hh1 = stage + 2 * H * H * 4;
        }
        __syncthreads();
        int cur = 0;
        for (int s = 0; s < CL_LSTM_LOOKBACK; ++s) {
            const int time = a.t - (CL_LSTM_LOOKBACK - 1) + s;
            const float* __restrict__ pre = g.gen_pre + ((long long)(time + row0) * a.n_bldg + b) * H * 4;
            const float xc = s == CL_LSTM_LOOKBACK - 1 ? cool_n : a.hist[(long long)(time % CL_LSTM_LOOKBACK) * plane + off];
            const float xt = a.hist[(long long)(CL_LSTM_LOOKBACK + (time - 1) % CL_LSTM_LOOKBACK) * plane + off];
            const float x2 = !two ? 0.0f : (s == CL_LSTM_LOOKBACK - 1 ? heat_n : a.hist[(long long)(2 * CL_LSTM_LOOKBACK + time % CL_LSTM_LOOKBACK) * plane + off]);
            float acc[CL_GEN_UPW][4];
#pragma unroll
            for (int j = 0; j < CL_GEN_UPW; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j][q] = 0.0f;
                if (j < n_u) {
                    const float* __restrict__ p4 = pre + (u0 + j) * 4;
                    const float* __restrict__ qx = wx + (u0 + j) * 12;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[j][q] = fmaf(qx[8 + q], x2, fmaf(qx[4 + q], xt, fmaf(qx[q], xc, p4[q])));      // (zero weights without a second input)
                }
            }
            lstm_gen_matvec<STAGED>(acc, m_hh0, h0b[cur], H, u0, n_u, lane);
            lstm_gen_update(acc, c0, h0b[cur ^ 1], u0, n_u, lane);
            __syncthreads();                                         // h0 of this window step complete
            if (layers == 2) {
#pragma unroll
                for (int j = 0; j < CL_GEN_UPW; ++j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[j][q] = j < n_u ? b1[(u0 + j) * 4 + q] : 0.0f;
                }
                lstm_gen_matvec<STAGED>(acc, m_ih1, h0b[cur ^ 1], H, u0, n_u, lane);
                lstm_gen_matvec<STAGED>(acc, m_hh1, h1b[cur], H, u0, n_u, lane);
                lstm_gen_update(acc, c1, h1b[cur ^ 1], u0, n_u, lane);
                __syncthreads();
            }
            cur ^= 1;
        }
        const float* top = layers == 2 ? h1b[cur] : h0b[cur];
        if (lead) {
            float out = W[CLW_BLIN];
            for (int u = 0; u < H; ++u) out = fmaf(wlin[u], top[u * 64 + lane], out);
            y = out;
            temp = fmaf(y, tmax - tmin, tmin);                                           // building.py:3031-3037
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < CL_GEN_UPW; ++j) {
                if (j < n_u) {
                    const int u = u0 + j;
                    hid[(long long)(0 * H + u) * a.n_env] = h0b[cur][u * 64 + lane]; hid[(long long)(1 * H + u) * a.n_env] = c0[j];
                    if (layers == 2) { hid[(long long)(2 * H + u) * a.n_env] = h1b[cur][u * 64 + lane]; hid[(long long)(3 * H + u) * a.n_env] = c1[j]; }
                }
            }
        }
    }
    if (live && lead) {
        a.hist[(long long)(CL_LSTM_LOOKBACK + slot) * plane + off] = y;              // building.py:3027-3028
        lstm_outputs(a, W, pre_t, off, plane, temp, cool, a.heat_dem ? a.heat_dem[off] : 0.0f);
    }
}

}  // namespace
#endif
