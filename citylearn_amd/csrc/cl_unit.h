// cl_unit.h -- one (env, building) unit advanced by one time step, fp32, gfx950.
//
// Device-side restatement of the reference's per-building step (paths relative to /root/reference/citylearn/):
//   Building.apply_actions            building.py:1500-1634   (order rules, inactive actions)
//   Building.update_*                 building.py:1641-1812   (devices, tanks, non-shiftable load, battery)
//   downward_electrical_flexibility   building.py:640-668     (power-outage coupling)
//   LSTMDynamicsBuilding.update_*_demand  building.py:3080-3158  (partial-load demand)
//   StorageDevice / StorageTank.charge    energy_model.py:719-768, 850-870
//   Battery.charge / curves / degrade     energy_model.py:1027-1141
//   Building.update_variables         building.py:2615-2703   (t = 0 repeat, net / cost / emission)
//
// Design notes (MI355X): every quantity that is identical for the 64 env lanes of a wavefront (building
// parameters, the time-series row of step t) lives in SGPRs (struct Bp / Row, filled by scalar loads);
// per-lane work is straight-line select/min/max/fma code -- the only divergent branches are the rare
// power-outage ordering cases.  Every division by a wave-uniform value was replaced on the host by a
// precomputed reciprocal / slope / intercept (float64, rounded once: the CLP_L_* block); the two per-lane
// transcendentals are the hardware v_rsq_f32 / v_rcp_f32 (1 ulp), far inside the 1e-4 parity bar.
//
// Two compile-time variants: LEAN (districts whose buildings only have battery + PV + non-shiftable load: the
// 2022 schemas and the headline benchmark) touches only the 32-word CLP_L_* parameter block and 5 row
// columns; FULL adds heat pump / heater / three tanks / outage / partial-load demand.
#pragma once

#include <stdint.h>
#include "../../include/citylearn_amd.h"

#ifdef CL_HOST_SHIM
// tests/host_shim compiles this header with g++ to single-step the unit arithmetic on a CPU-only box while
// debugging numerics.  It is a test harness, never a product path (the library has no CPU backend).
#include <math.h>
#include <string.h>
#define CL_DEV inline
namespace cl {
inline float rcp(float x) { return 1.0f / x; }
inline float rsq(float x) { return 1.0f / sqrtf(x); }
inline float fsqrt(float x) { return sqrtf(x); }
inline float med3(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline float __powf(float a, float b) { return powf(a, b); }
}
#else
#include <hip/hip_runtime.h>
#define CL_DEV __device__ __forceinline__
namespace cl {
CL_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
CL_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
CL_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
CL_DEV float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
}
#endif
#define CL_ZDP 1e-6f          // data.py:19 ZERO_DIVISION_PLACEHOLDER
// Place a wave-uniform value in a VGPR once and keep it there.  A VALU instruction reads at most one SGPR, so selecting between two
// parameters (`lo ? b0 : b1`) costs a v_mov per use; the compiler re-materialises that copy for every env of a lane instead of
// keeping it, because a copy is "free" to recompute.
#ifdef CL_HOST_SHIM
#define CL_PIN_V(x) ((void)0)
#else
#define CL_PIN_V(x) asm("" : "+v"(x))
#endif

namespace cl {

// Per-building parameters, wave-uniform (SGPRs).  They are loaded in groups right where they are used (each group
// sits in its own basic block behind a wave-uniform flag test) so that the general kernel never holds all ~90
// parameter words at once: SGPR pressure, not arithmetic, was what limited its occupancy.
struct BattP {      // battery block of the CLP_L_* words
    float r, pdt, pow, cap, capl, inv_cap, inv_pow, omd, degk;
    float cpc_x1, cpc_a0, cpc_b0, cpc_a1, cpc_b1;
    float pec_x1, pec_x2, pec_x3, pec_a0, pec_b0, pec_a1, pec_b1, pec_a2, pec_b2, pec_a3, pec_b3;
};
struct TankP { float cap, capl, rte, irte, icap, maxin, maxout; };
struct Bp {         // what stays live for the whole unit
    const uint32_t* __restrict__ p;
    uint32_t flags;
    int a_es;
    float r, rw_exponent;
    BattP batt;     // lean kernel only (loaded up front); the general kernel loads it at the battery call site
    // ---- general kernel head ----
    float dt, cd_pow, hd_pow, dd_pow, t0_iheat_div, dyn_warmup;
    int a_cs, a_hs, a_ds, a_cd, a_hd, a_coh;
};

CL_DEV float pw(const uint32_t* __restrict__ p, int slot) { return __uint_as_float(p[slot]); }

// a * b that is never fused into a following add: the district cost is a sum of rounded per-building products (citylearn.py:1909-1918),
// and whether `q += net * price` became an fma used to depend on the instantiation (one vs two vs four envs per lane).
CL_DEV float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

CL_DEV void load_batt(BattP& B, const uint32_t* __restrict__ p) {
    B.r = pw(p, CLP_L_TSR); B.pdt = pw(p, CLP_L_PDT); B.pow = pw(p, CLP_L_POW); B.cap = pw(p, CLP_L_CAP);
    B.capl = pw(p, CLP_L_CAPL); B.inv_cap = pw(p, CLP_L_INV_CAP); B.inv_pow = pw(p, CLP_L_INV_POW);
    B.omd = pw(p, CLP_L_OMD); B.degk = pw(p, CLP_L_DEGK);
    B.cpc_x1 = pw(p, CLP_L_CPC_X1); B.cpc_a0 = pw(p, CLP_L_CPC_A0); B.cpc_b0 = pw(p, CLP_L_CPC_B0);
    B.cpc_a1 = pw(p, CLP_L_CPC_A1); B.cpc_b1 = pw(p, CLP_L_CPC_B1);
    B.pec_x1 = pw(p, CLP_L_PEC_X1); B.pec_x2 = pw(p, CLP_L_PEC_X2); B.pec_x3 = pw(p, CLP_L_PEC_X3);
    B.pec_a0 = pw(p, CLP_L_PEC_A0); B.pec_b0 = pw(p, CLP_L_PEC_B0); B.pec_a1 = pw(p, CLP_L_PEC_A1);
    B.pec_b1 = pw(p, CLP_L_PEC_B1); B.pec_a2 = pw(p, CLP_L_PEC_A2); B.pec_b2 = pw(p, CLP_L_PEC_B2);
    B.pec_a3 = pw(p, CLP_L_PEC_A3); B.pec_b3 = pw(p, CLP_L_PEC_B3);
}

CL_DEV void load_tank(TankP& T, const uint32_t* __restrict__ p, int raw, int der) {
    T.cap = pw(p, raw); T.rte = pw(p, raw + 2); T.maxin = pw(p, raw + 4); T.maxout = pw(p, raw + 5);
    T.irte = pw(p, der); T.icap = pw(p, der + 1); T.capl = pw(p, der + 2);
}

template <bool FULL>
CL_DEV void load_bp(Bp& B, const uint32_t* __restrict__ p) {
    B.p = p;
    B.flags = p[CLP_L_FLAGS]; B.a_es = (int)p[CLP_L_ACT_ES];
    B.r = pw(p, CLP_L_TSR); B.rw_exponent = pw(p, CLP_L_RW_EXPONENT);
    if constexpr (!FULL) {
        load_batt(B.batt, p);
    } else {
        B.dt = pw(p, CLP_DT_HOURS);
        B.cd_pow = pw(p, CLP_CD_POW); B.hd_pow = pw(p, CLP_HD_POW); B.dd_pow = pw(p, CLP_DD_POW);
        B.t0_iheat_div = pw(p, CLP_T0_IHEAT_DIV); B.dyn_warmup = pw(p, CLP_DYN_WARMUP);
        B.a_cs = (int)p[CLP_ACT_COOL_STO]; B.a_hs = (int)p[CLP_ACT_HEAT_STO]; B.a_ds = (int)p[CLP_ACT_DHW_STO];
        B.a_cd = (int)p[CLP_ACT_COOL_DEV]; B.a_hd = (int)p[CLP_ACT_HEAT_DEV]; B.a_coh = (int)p[CLP_ACT_COH_DEV];
    }
}

// Time-series row of (t, building): wave-uniform.
// (Field order: values that sit next to each other in the table -- load / solar, price / carbon -- are kept apart here.  With them
//  adjacent the optimiser kept a 16-byte stack slot of the struct alive in the two-envs-per-lane kernel (vector-typed pair
//  accesses it could not promote): a scratch store and two scratch loads per building.)
struct Row {
    float nsl, cool, sol, heat, price, dhw, carbon, cop_c, cop_h, cop_d, icop_c, icop_h, icop_d, hvac;
    float icop_h_eval;      // 1 / heating COP on the episode's LAST row: what evaluate()'s partial-load baseline divides by (sic, building.py:2893-2898)
    bool outage;
};

// `q_eval`: the same building's row at the episode's last step (only its heating COP is read; nullptr = this row)
template <bool FULL>
CL_DEV void load_row(Row& R, const float* __restrict__ q, uint32_t flags, const float* __restrict__ q_eval = nullptr) {
    R.nsl = q[CLT_NSL]; R.sol = q[CLT_SOLAR]; R.price = q[CLT_PRICE]; R.carbon = q[CLT_CARBON];
    R.outage = false;
    if constexpr (FULL) {
        R.icop_h_eval = (q_eval ? q_eval : q)[CLT_ICOP_HEAT];
        R.cool = q[CLT_COOL_DEM]; R.heat = q[CLT_HEAT_DEM]; R.dhw = q[CLT_DHW_DEM];
        R.cop_c = q[CLT_COP_COOL]; R.cop_h = q[CLT_COP_HEAT]; R.cop_d = q[CLT_COP_DHW];
        R.icop_c = q[CLT_ICOP_COOL]; R.icop_h = q[CLT_ICOP_HEAT]; R.icop_d = q[CLT_ICOP_DHW];
        R.hvac = q[CLT_HVAC_MODE];
        R.outage = (flags & CLF_OUTAGE) && (q[CLT_OUTAGE] != 0.0f);
    }
}

// The same row through 32-bit integer loads: float stores to the state / output planes cannot alias them (type-based alias
// analysis), so inside a loop that also stores the row still arrives by scalar loads in SGPRs instead of 14 uniform VGPRs.
// `ts` is read-only for every kernel, which is what makes the pun harmless.
template <bool FULL>
CL_DEV void load_row_scalar(Row& R, const float* __restrict__ qf, uint32_t flags, const float* __restrict__ qf_eval = nullptr) {
    const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(qf);
    if constexpr (FULL) R.icop_h_eval = pw(reinterpret_cast<const uint32_t*>(qf_eval ? qf_eval : qf), CLT_ICOP_HEAT);
    R.nsl = pw(q, CLT_NSL); R.sol = pw(q, CLT_SOLAR); R.price = pw(q, CLT_PRICE); R.carbon = pw(q, CLT_CARBON);
    R.outage = false;
    if constexpr (FULL) {
        R.cool = pw(q, CLT_COOL_DEM); R.heat = pw(q, CLT_HEAT_DEM); R.dhw = pw(q, CLT_DHW_DEM);
        R.cop_c = pw(q, CLT_COP_COOL); R.cop_h = pw(q, CLT_COP_HEAT); R.cop_d = pw(q, CLT_COP_DHW);
        R.icop_c = pw(q, CLT_ICOP_COOL); R.icop_h = pw(q, CLT_ICOP_HEAT); R.icop_d = pw(q, CLT_ICOP_DHW);
        R.hvac = pw(q, CLT_HVAC_MODE);
        R.outage = (flags & CLF_OUTAGE) && (pw(q, CLT_OUTAGE) != 0.0f);
    }
}

// Carried per-unit state (one lane).  eff_lo / deg_lo: CLD_F64_MAPS only -- the low words of Battery.efficiency and
// Battery.degraded_capacity, which the reference carries as float64 (value = (double)hi + (double)lo).
struct State { float soc, eff, degcap, cs, hs, ds, eff_lo, deg_lo; };
// Actions of one unit (inactive -> 0 for storages / ignored for devices, building.py:1557-1564).
struct Act { float cs, hs, ds, es, cd, hd; };
// Per-unit results of the step.
struct Out { float net, cost, emission, eb, cool_dem, heat_dem, dhw_dem, c_cool, c_heat, c_dhw, c_ns, base_net, expected, served, net_ws, se_cool, se_heat, se_dhw;; uint32_t viol; };
// (`viol`: CLV_* bits of the reference's runtime assertions this unit would have tripped -- written by unit_step<.., CHECK = true> only, CLD_CHECK)
// Running electricity_consumption[t] of the five electric devices.
struct Acc { float c_cool, c_heat, c_dhw, c_ns, c_b; };

// CLD_CHECK (a debug mode: one more output word per unit): the reference's own runtime assertions, evaluated where the reference evaluates
// them, with its tolerance (data.py:18 TOLERANCE = 1e-4) -- `viol` collects CLV_* bits instead of raising, the host raises (CityLearnEnv.step)
#define CL_TOLERANCE 1e-4f
template <bool CHECK = false>
CL_DEV float flexibility(const Bp& B, const Row& R, const Acc& A, [[maybe_unused]] uint32_t* viol = nullptr) {
    // building.py:640-668 (+inf unless this (t, building) is in a power outage)
    if (!R.outage) return INFINITY;
    const float used = (A.c_cool + A.c_heat + A.c_dhw + A.c_ns + A.c_b) * B.r;
    const float capacity = fabsf(R.sol) - used;
    if constexpr (CHECK) {
        if (!(capacity >= 0.0f || fabsf(capacity) < CL_TOLERANCE)) *viol |= CLV_FLEXIBILITY;      // building.py:665
    }
    return fmaxf(0.0f, capacity);
}

// Battery.charge (energy_model.py:1027-1057) on top of update_electrical_storage (building.py:1791-1812).
// `flex` is downward_electrical_flexibility at the moment of the call; the battery's own
// electricity_consumption[t] is still 0 there (one charge() per step), so available_nominal_power == P.
CL_DEV float battery_energy(const BattP& B, float E, State& S);
CL_DEV float battery_step(const BattP& B, float a_es, float flex, State& S) {
    return battery_energy(B, fminf(a_es * B.pdt, flex), S);
}

// Battery.charge(E) for an energy E [kWh] that already carries Battery.charge's own `* time_step_ratio`
// (energy_model.py:1036); also the entry point of the EV batteries (cl_flex.h).
CL_DEV float battery_energy(const BattP& B, float E, State& S) {
    const float prev = S.soc;
    const float e_init = fmaxf(0.0f, prev * B.capl);                                    // energy_model.py:661-666
    const float socn = e_init * B.inv_cap;
    // capacity_power_curve (energy_model.py:1070-1090), slopes/intercepts pre-multiplied by nominal_power
    // (fields are copied to locals first: selecting between struct members through a reference makes LLVM
    //  select the *addresses* and parks the whole struct in scratch memory)
    const float ca0 = B.cpc_a0, cb0 = B.cpc_b0, ca1 = B.cpc_a1, cb1 = B.cpc_b1;
    const float pa0 = B.pec_a0, pb0 = B.pec_b0, pa1 = B.pec_a1, pb1 = B.pec_b1, pa2 = B.pec_a2, pb2 = B.pec_b2,
                pa3 = B.pec_a3, pb3 = B.pec_b3;
    const bool lo = socn <= B.cpc_x1;
    const float pmax = fmaf(lo ? cb0 : cb1, socn, lo ? ca0 : ca1);
    // charge: min(pmax, P, degraded - e_init, E); discharge: max(-pmax, -DoD limit, E)   (1036-1050)
    const float e_chg = fminf(fminf(pmax, B.pow), fminf(S.degcap - e_init, E));
    const float lim = -fmaxf((prev - B.omd) * B.cap * fsqrt(S.eff), 0.0f);              // previous call's efficiency
    const float e_dis = fmaxf(fmaxf(-pmax, lim), E);
    float e = E >= 0.0f ? e_chg : e_dis;
    // power_efficiency_curve at min(|E|, pmax)/P (energy_model.py:1039, 1052, 1092-1109)
    const float x = fabsf(fminf(fabsf(E), pmax)) * B.inv_pow;
    const bool s0 = x <= B.pec_x1, s1 = x <= B.pec_x2, s2 = x <= B.pec_x3;
    const float eb_ = s0 ? pb0 : s1 ? pb1 : s2 ? pb2 : pb3;
    const float ea_ = s0 ? pa0 : s1 ? pa1 : s2 ? pa2 : pa3;
    const float eff = fmaf(eb_, x, ea_);
    const float irte = rsq(eff), rte = eff * irte;
    // StorageDevice.charge with the nominal capacity (energy_model.py:719-768)
    e *= B.r;
    // charge: min(e_init + e rte, cap); discharge: max(0, e_init + e / rte).  0 <= e_init <= cap, so the unused bound of
    // either branch is inactive and both collapse into one clamp of one fma (same bits, three instructions fewer per unit)
    const bool chg = e >= 0.0f;
    const float e_fin = med3(fmaf(e, chg ? rte : irte, e_init), 0.0f, B.cap);
    const float d = e_fin - e_init;                                                      // d >= 0 exactly when charging (or d == 0)
    const float eb = d * (chg ? irte : rte);
    // degrade with the pre-step degraded capacity (energy_model.py:1130-1141)
    S.degcap = fmaxf(S.degcap - B.degk * fabsf(eb) * rcp(fmaxf(S.degcap, CL_ZDP)), 0.0f);
    S.eff = eff;
    S.soc = e_fin * B.inv_cap;
    return eb;
}

// ---- CLD_F64_MAPS: the battery map in the reference's own precision model ---------------------------------------------------
// The reference evaluates Battery.charge with Python floats and numpy scalars and stores soc[t] / energy_balance[t] in float32
// series (energy_model.py:1027-1141).  Under numpy's promotion rules (NEP 50: a Python float is "weak", so float32 op python-float
// stays float32, while float32 op np.float64 -- curve tables, time_step_ratio -- becomes float64) that is MOSTLY float64 with a few
// float32 operations in fixed places: `prev_soc * capacity`, the discharge limit `(prev_soc - soc_limit) * capacity`, and
// `capacity_loss_coefficient * capacity * |energy_balance|`.  The map soc_t = f(soc_t-1, a_t) is locally expansive on the steep
// part of the capacity-power curve, so the ~1 ulp by which an all-fp32 evaluation differs can grow to ~2e-4 relative over a few
// steps of a free-running episode.  This variant removes the seed instead of bounding the growth: it follows
// oracle/oracle.py (bit-exact against the reference) operation by operation and type by type -- the same divisions, the same
// roundings, on the packer's unrounded float64 parameters (CLP_D_* block) -- with efficiency and degraded capacity, which the
// reference carries as float64 attributes, held as hi + lo float32 planes (48 bits).  On the first charge() of an episode the two
// attributes are still Python floats in the reference, which moves two more operations to float32 (`first`).
struct BattP64 {
    double r, dt, pow, cap, oml /* 1 - loss_coefficient r */, soc_limit /* 1 - depth_of_discharge */, clccap /* capacity_loss_coefficient * capacity */;
    double cx[3], cy[3], px[5], py[5];      // capacity_power_curve, power_efficiency_curve (x, y)
    double rcap, rpow, rcx[2], rpx[4];      // correctly rounded reciprocals of the constant divisors (CLPD_RCAP ..): div_rn below
};

// RN(a / b) for a divisor whose correctly rounded reciprocal rb = RN(1 / b) is at hand: q0 = RN(a rb) is within two ulps of the quotient,
// one residual step (the residual a - b q is exact in a fused multiply-add) brings it within a fraction of an ulp, and Markstein's theorem
// (IBM J. Res. Dev. 34, 1990) makes the second step's result the correctly rounded quotient -- the bits IEEE division returns, which is
// what the reference computes.  Five FMA-class operations; a non-finite reciprocal (degenerate curve segment) takes the hardware division.
CL_DEV double div_rn(double a, double b, double rb) {
    if (!(fabs(rb) < 1.0e300)) return a / b;                   // inf / nan reciprocal (uniform: a parameter of the building)
    const double q0 = a * rb;
    const double q1 = __builtin_fma(__builtin_fma(-q0, b, a), rb, q0);
    return __builtin_fma(__builtin_fma(-q1, b, a), rb, q1);
}

CL_DEV double pd(const uint32_t* __restrict__ p, int k) {           // k-th double of the CLP_D_* block (8-byte aligned: CL_NP and CLP_D_FIRST are even)
    const uint64_t bits = (uint64_t)p[CLP_D_FIRST + 2 * k] | ((uint64_t)p[CLP_D_FIRST + 2 * k + 1] << 32);
    double d;
    __builtin_memcpy(&d, &bits, 8);
    return d;
}

CL_DEV void load_batt64(BattP64& B, const uint32_t* __restrict__ p) {
    B.r = pd(p, CLPD_TSR); B.dt = pd(p, CLPD_DT); B.pow = pd(p, CLPD_POW); B.cap = pd(p, CLPD_CAP); B.oml = pd(p, CLPD_OML);
    B.soc_limit = pd(p, CLPD_SOC_LIMIT); B.clccap = pd(p, CLPD_CLCCAP);
#pragma unroll
    for (int k = 0; k < 3; ++k) { B.cx[k] = pd(p, CLPD_CPC_X0 + k); B.cy[k] = pd(p, CLPD_CPC_Y0 + k); }
#pragma unroll
    for (int k = 0; k < 5; ++k) { B.px[k] = pd(p, CLPD_PEC_X0 + k); B.py[k] = pd(p, CLPD_PEC_Y0 + k); }
    B.rcap = pd(p, CLPD_RCAP); B.rpow = pd(p, CLPD_RPOW);
    B.rcx[0] = pd(p, CLPD_RCPC_01); B.rcx[1] = pd(p, CLPD_RCPC_12);
#pragma unroll
    for (int k = 0; k < 4; ++k) B.rpx[k] = pd(p, CLPD_RPEC_01 + k);
}

// Battery.charge(energy): `energy` [kWh] is what Building.update_electrical_storage hands over (building.py:1801-1812), float64.
// Returns energy_balance[t] as the float32 the reference stores.  `first`: the first charge() since reset (step 0).
CL_DEV float battery_charge_ref(const BattP64& B, double energy, bool first, State& S) {
#pragma clang fp contract(off)                                                            // the reference rounds every product before it adds
    const double ZDP = 1e-6;                                                              // data.py:19
    const float prev = S.soc;                                                             // soc[t-1]: a float32 series value
    const double eff_prev = (double)S.eff + (double)S.eff_lo;
    const double degcap = (double)S.degcap + (double)S.deg_lo;
    // (fields copied to locals first: a select between struct members through a reference becomes a select between their
    //  addresses and parks the whole struct in scratch memory)
    const double cx1 = B.cx[1], cx2 = B.cx[2], cx0 = B.cx[0], cy0 = B.cy[0], cy1 = B.cy[1], cy2 = B.cy[2];
    const double px0 = B.px[0], px1 = B.px[1], px2 = B.px[2], px3 = B.px[3], px4 = B.px[4];
    const double py0 = B.py[0], py1 = B.py[1], py2 = B.py[2], py3 = B.py[3], py4 = B.py[4];
    energy = energy * B.r;                                                                // energy_model.py:1036
    const double action_energy = energy;
    // energy_init (661-666): float32 product `prev_soc * capacity`, then float64 `* (1 - loss_coefficient)`
    const double e_init = fmax(0.0, (double)(prev * (float)B.cap) * B.oml);
    // get_max_input_power (1070-1090): idx = max(0, argmax(soc <= xs) - 1) -- 0 again when no breakpoint is >= soc
    const double capz = fmax(B.cap, ZDP), powz = fmax(B.pow, ZDP);
    const double socn = div_rn(e_init, capz, B.rcap);
    const bool seg1 = !(socn <= cx1) && (socn <= cx2);
    const double xa = seg1 ? cx1 : cx0, xb = seg1 ? cx2 : cx1, ya = seg1 ? cy1 : cy0, yb = seg1 ? cy2 : cy1;
    const double rc0 = B.rcx[0], rc1 = B.rcx[1];
    const double pmax = B.pow * (ya + div_rn((yb - ya) * (socn - xa), xb - xa, seg1 ? rc1 : rc0));
    double e_eff;                                                                         // argument of get_current_efficiency
    if (energy >= 0.0) {
        const double wrt_degrade = degcap - e_init;
        energy = fmin(fmin(pmax, B.pow), fmin(wrt_degrade, energy));                      // (the battery's own consumption[t] is still 0: one charge() per step)
        e_eff = fmin(action_energy, pmax);
    } else {
        // limit = max((prev_soc - soc_limit) * capacity * rte, 0) * -1: float32 up to `* rte`, which is float32 too while the
        // efficiency is still the Python float of reset() and float64 afterwards (rte = efficiency ** 0.5 of the PREVIOUS call)
        const float l32 = (prev - (float)B.soc_limit) * (float)B.cap;
        const double rte_prev = sqrt(eff_prev);
        const double lim = first ? (double)(l32 * (float)rte_prev) : (double)l32 * rte_prev;
        const double limit = -fmax(lim, 0.0);
        energy = fmax(fmax(-pmax, limit), energy);
        e_eff = fmin(fabs(action_energy), pmax);
    }
    // get_current_efficiency (1092-1109)
    const double x = div_rn(fabs(e_eff), powz, B.rpow);
    const int seg = (x <= px1) ? 0 : (x <= px2) ? 1 : (x <= px3) ? 2 : (x <= px4) ? 3 : 0;
    const double qa = seg == 0 ? px0 : seg == 1 ? px1 : seg == 2 ? px2 : px3, qb = seg == 0 ? px1 : seg == 1 ? px2 : seg == 2 ? px3 : px4;
    const double ra = seg == 0 ? py0 : seg == 1 ? py1 : seg == 2 ? py2 : py3, rb = seg == 0 ? py1 : seg == 1 ? py2 : seg == 2 ? py3 : py4;
    const double rp0 = B.rpx[0], rp1 = B.rpx[1], rp2 = B.rpx[2], rp3 = B.rpx[3];
    const double eff = ra + div_rn((x - qa) * (rb - ra), qb - qa, seg == 0 ? rp0 : seg == 1 ? rp1 : seg == 2 ? rp2 : rp3);
    const double rte = sqrt(eff);
    // StorageDevice.charge (719-768).  The reference divides by the round-trip efficiency once per call: `energy / rte` when discharging
    // (to get e_fin), `d / rte` when charging (energy_balance from the stored difference) -- one hardware division on the numerator the
    // call's sign selects (as two selects between a product and a quotient it was two divisions on every lane).
    energy = energy * B.r;
    const bool charging = energy >= 0.0;
    const double e_fin_c = fmin(e_init + energy * rte, B.cap);
    const double quot = (charging ? e_fin_c - e_init : energy) / rte;
    const double e_fin = charging ? e_fin_c : fmax(0.0, e_init + quot);
    const float soc = (float)div_rn(e_fin, capz, B.rcap);                                 // soc[t]: float32 series
    const double d = e_fin - e_init;
    // (charging: d = e_fin_c - e_init >= 0 and quot = d / rte; discharging: e_fin <= e_init, so d <= 0 -> d * rte, and d == 0 gives 0 either way)
    const float eb = (float)(charging && d >= 0.0 ? quot : d * rte);                     // energy_balance[t]: float32 series
    // degrade (1130-1141): float32 `clc * capacity * |eb|`; the division is float32 too while degraded_capacity is still a Python float
    const float g32 = (float)B.clccap * fabsf(eb);
    const double den = 2.0 * fmax(degcap, ZDP);
    const double deg = (first ? (double)(g32 / (float)den) : (double)g32 / den) * B.r;
    const double deg_new = fmax(degcap - deg, 0.0);
    S.soc = soc;
    S.eff = (float)eff; S.eff_lo = (float)(eff - (double)S.eff);
    S.degcap = (float)deg_new; S.deg_lo = (float)(deg_new - (double)S.degcap);
    return eb;
}

// ---- CLD_F64_CHAIN: the soc chain in float64, everything else as in the fp32 map ---------------------------------------------------
// What makes the all-fp32 map drift on free-running episodes is not its arithmetic but ONE rounding: the degraded capacity, a float64
// attribute of the reference's Battery, kept as a float32 plane.  Whenever the charge clamp `degraded_capacity - energy_init` binds, that
// half-ulp lands in soc[t] (d soc / d degraded_capacity = round-trip efficiency / capacity), and the steep segment of the capacity-power
// curve -- where the battery then sits -- multiplies a soc error by 1 - |slope| rte / capacity when charging and 1 + |slope| / (rte
// capacity) when discharging (2022 batteries: -1.6 and +3.9 per step).  tests/test_f64_maps_host.py measures it: fp32 arithmetic with the
// degraded capacity exact -- 11 x (2022) / 2.2 x (2020) less soc drift; float64 arithmetic with a float32 degraded capacity -- no gain at
// all; float64 chain + exact degraded capacity -- 70 - 170 x less, soc[t] equal to the reference's float32 value at ~90 % of all steps.
// So this variant (a) carries the LOSS D = capacity - degraded_capacity in the float32 plane (D << capacity: its half-ulp is ~2^-40 of
// the capacity; the degradation increments, ~1e-5 kWh, are added in fp32) and (b) evaluates energy_init -> capacity-power limit ->
// selected energy -> efficiency -> round-trip efficiency -> final energy -> soc[t] / energy_balance[t] in float64 -- rounded to
// float32 exactly where the reference's float32 series round -- on host-prepared constants (CLP_C_*): both curves as a first segment
// plus one ramp per breakpoint (no segment selection: a float64 select is two v_cndmask), reciprocals instead of divisions, 1 / sqrt
// from the fp32 seed and one Newton step (2^-45).  The efficiency of the previous call stays a float32 plane: it only enters the
// discharge limit, which ends the step at (nearly) zero energy whatever came before.
struct BattC {
    double cap, oml, rcap, pdt, pow, rpow, r, ca0, cb0, cx1, cdb1, ea0, eb0, ex1, edb1, ex2, edb2, ex3, edb3;
    float cap32, omd32, degk;
};

CL_DEV double pc(const uint32_t* __restrict__ p, int k) {           // k-th double of the CLP_C_* block (8-byte aligned)
    const uint64_t bits = (uint64_t)p[CLP_C_FIRST + 2 * k] | ((uint64_t)p[CLP_C_FIRST + 2 * k + 1] << 32);
    double d;
    __builtin_memcpy(&d, &bits, 8);
    return d;
}

// the float64 part, from the CLP_C_* block at p + CLP_C_FIRST
CL_DEV void load_battc64(BattC& C, const uint32_t* __restrict__ p) {
    C.cap = pc(p, CLPC_CAP); C.oml = pc(p, CLPC_OML); C.rcap = pc(p, CLPC_RCAP); C.pdt = pc(p, CLPC_PDT); C.pow = pc(p, CLPC_POW);
    C.rpow = pc(p, CLPC_RPOW); C.r = pc(p, CLPC_TSR);
    C.ca0 = pc(p, CLPC_CPC_A0); C.cb0 = pc(p, CLPC_CPC_B0); C.cx1 = pc(p, CLPC_CPC_X1); C.cdb1 = pc(p, CLPC_CPC_DB1);
    C.ea0 = pc(p, CLPC_PEC_A0); C.eb0 = pc(p, CLPC_PEC_B0); C.ex1 = pc(p, CLPC_PEC_X1); C.edb1 = pc(p, CLPC_PEC_DB1);
    C.ex2 = pc(p, CLPC_PEC_X2); C.edb2 = pc(p, CLPC_PEC_DB2); C.ex3 = pc(p, CLPC_PEC_X3); C.edb3 = pc(p, CLPC_PEC_DB3);
}

CL_DEV void load_battc(BattC& C, const uint32_t* __restrict__ p) {
    load_battc64(C, p);
    C.cap32 = pw(p, CLP_L_CAP); C.omd32 = pw(p, CLP_L_OMD); C.degk = pw(p, CLP_L_DEGK);
}

// 1 / sqrt(x) to ~2^-45: the fp32 seed (v_rsq_f32, 1 ulp) and one Newton step whose residual is a single fused multiply-add
CL_DEV double rsq64(double x) {
    const double y = (double)rsq((float)x);
    const double e = __builtin_fma(-(0.5 * x) * y, y, 0.5);
    return __builtin_fma(y, e, y);
}

// update_electrical_storage + Battery.charge (building.py:1801-1812, energy_model.py:1027-1141) for one unit: `a_es` the action, `flex` the
// downward electrical flexibility [kWh] at the moment of the call.  S.degcap holds the capacity LOSS.  Returns energy_balance[t].
CL_DEV float battery_charge_chain(const BattC& C, float a_es, float flex, State& S) {
    const float prev = S.soc;
    // energy_init (661-666): the reference's float32 product `prev_soc * capacity`, then float64
    const double e_init = fmax(0.0, (double)(prev * C.cap32) * C.oml);
    const double socn = e_init * C.rcap;
    const double pmax = __builtin_fma(C.cdb1, fmax(socn - C.cx1, 0.0), __builtin_fma(C.cb0, socn, C.ca0));                // (1070-1090)
    const double E = fmin((double)a_es * C.pdt, (double)flex);                          // energy handed to charge(): `/ r` there, `* r` inside (1036)
    const double degcap = C.cap - (double)S.degcap;
    // charge: min(pmax, P, degraded - e_init, E); discharge: max(-pmax, -DoD limit, E) (1036-1050); the limit from float32 operands, like the reference
    const double e_chg = fmin(fmin(pmax, C.pow), fmin(degcap - e_init, E));
    const double lim = fmax((double)((prev - C.omd32) * C.cap32 * fsqrt(S.eff)), 0.0);
    const double e_dis = fmax(fmax(-pmax, -lim), E);
    const bool chg = E >= 0.0;
    double e = chg ? e_chg : e_dis;
    // power_efficiency_curve at min(|E|, pmax) / P (1039, 1052, 1092-1109)
    const double x = fmin(fabs(E), pmax) * C.rpow;
    double eff = __builtin_fma(C.eb0, x, C.ea0);
    eff = __builtin_fma(C.edb1, fmax(x - C.ex1, 0.0), eff);
    eff = __builtin_fma(C.edb2, fmax(x - C.ex2, 0.0), eff);
    eff = __builtin_fma(C.edb3, fmax(x - C.ex3, 0.0), eff);
    const double irte = rsq64(eff), rte = eff * irte;
    // StorageDevice.charge (719-768): one clamp of one fma, as in the fp32 map (0 <= e_init <= capacity)
    // (the branch of StorageDevice.charge follows the sign of the CLAMPED energy, as in battery_energy / battery_charge_ref: a battery sitting at
    //  its degraded capacity has `degraded - e_init` slightly negative after the degradation step, and the reference then discharges with
    //  energy / rte although charging was requested -- round-5 advisor finding)
    e *= C.r;
    const bool chg_e = e >= 0.0;
    const double e_fin = fmin(fmax(__builtin_fma(e, chg_e ? rte : irte, e_init), 0.0), C.cap);
    const double d = e_fin - e_init;
    const float eb = (float)(d * (chg_e ? irte : rte));                                  // energy_balance[t]: float32 series
    // degrade (1130-1141) on the pre-step degraded capacity: the increment (~1e-5 kWh) in fp32, accumulated into the loss
    const float degcap32 = C.cap32 - S.degcap;
    S.degcap = fminf(fmaf(C.degk * fabsf(eb), rcp(fmaxf(degcap32, CL_ZDP)), S.degcap), C.cap32);
    S.eff = (float)eff;
    S.soc = (float)(e_fin * C.rcap);                                                     // soc[t]: float32 series
    return eb;
}

// StorageDevice.charge under StorageTank.charge's power clamps (energy_model.py:719-768, 850-870).
// `e` is the energy handed to tank.charge() after `_convert_energy_for_storage` (building.py:1814-1823),
// i.e. it is multiplied by time_step_ratio twice on its way in.
CL_DEV void tank_charge(float e, float prev_soc, float cap, float capl, float rte, float irte, float icap,
                        float maxin, float maxout, float r, float& soc, float& eb) {
    e *= r;
    e = e >= 0.0f ? fminf(e, maxin) : fmaxf(-maxout, e);
    e *= r;
    const float e_init = fmaxf(0.0f, prev_soc * capl);
    const float e_fin = e >= 0.0f ? fminf(fmaf(e, rte, e_init), cap) : fmaxf(0.0f, fmaf(e, irte, e_init));
    soc = e_fin * icap;
    const float d = e_fin - e_init;
    eb = d * (d >= 0.0f ? irte : rte);
}

// One end use (cooling / heating / dhw): device + storage in the order given by the storage action's sign.
template <bool CHECK = false>
CL_DEV void end_use(const Bp& B, const Row& R, Acc& A, float& c, float demand, float a_sto, float cscale,
                    float dev_pow, float cop, float icop, const TankP& T, float ir, float& soc, float& eb, float& e_dev,
                    [[maybe_unused]] uint32_t* viol = nullptr, [[maybe_unused]] uint32_t polarity_bit = 0u) {
    const float prev_soc = soc;
    const float energy = a_sto * cscale;
    const bool disc = a_sto < 0.0f;                         // storage first (building.py:1611-1622)
    // storage-first lanes discharge now; the others see an untouched tank (energy_balance[t] == 0)
    float soc_a, eb_a;
    tank_charge(fmaxf(-demand, energy) * ir, prev_soc, T.cap, T.capl, T.rte, T.irte, T.icap, T.maxin, T.maxout, B.r, soc_a, eb_a);
    eb_a = disc ? eb_a : 0.0f;
    // device (building.py:1641-1661): `c` is this end use's accumulator inside A
    float max_out = fminf(flexibility<CHECK>(B, R, A, viol), dev_pow - c * B.r) * cop;
    const float out = fminf(demand - fmaxf(-eb_a, 0.0f), max_out);
    e_dev = out;
    if constexpr (CHECK) {
        // ___electricity_consumption_polarity_check (building.py:1831-1835, called at 1660 / 1708 / 1753)
        const float consumption = out * icop;
        if (!(consumption >= 0.0f || fabsf(consumption) < CL_TOLERANCE)) *viol |= polarity_bit;
    }
    c += fmaxf(0.0f, out * icop);
    // storage after the device (charging or idle lanes; building.py:1663-1687; the charging branch reads the flexibility again)
    max_out = fminf(flexibility<CHECK>(B, R, A, viol), dev_pow - c * B.r) * cop;
    const float e_c = energy > 0.0f ? fminf(max_out, energy) : fmaxf(-demand, energy);
    float soc_c, eb_c;
    tank_charge(e_c * ir, prev_soc, T.cap, T.capl, T.rte, T.irte, T.icap, T.maxin, T.maxout, B.r, soc_c, eb_c);
    soc = disc ? soc_a : soc_c;
    eb = disc ? eb_a : eb_c;
    c += fmaxf(eb, 0.0f) * icop;
}

// update_electrical_storage (building.py:1801-1812) + Battery.charge in the reference's precision model (CLD_F64_MAPS):
// power = action * nominal_power; energy = power * dt; energy = min(energy, flexibility); charge(energy / time_step_ratio).
CL_DEV float battery_step_f64(const uint32_t* __restrict__ p, float a_es, float flex, bool first, State& S) {
#pragma clang fp contract(off)
    BattP64 bp;
    load_batt64(bp, p);
    const double energy = fmin((double)a_es * bp.pow * bp.dt, (double)flex);
    return battery_charge_ref(bp, energy / bp.r, first, S);
}

// The whole unit step.  `t` and `t0_quirk` are wave-uniform.  F64: the battery map in float64 (CLD_F64_MAPS).
// PREC: 0 = fp32 battery map, 1 = CLD_F64_MAPS (battery_charge_ref), 2 = CLD_F64_CHAIN (battery_charge_chain; S.degcap is the capacity loss)
// CHECK (CLD_CHECK): O.viol = the CLV_* bits of the reference assertions the unit tripped (building.py:665, 1831-1835; energy_model.py:146-148)
template <bool FULL, int PREC = 0, bool CHECK = false>
CL_DEV void unit_step(const Bp& B, const Row& R, int t, bool t0_quirk, const Act& a, State& S, Out& O) {
    constexpr bool F64 = PREC == 1;
    [[maybe_unused]] uint32_t viol = 0u;
    const bool first = t0_quirk && t == 0;
    const bool has_batt = B.flags & CLF_BATTERY;
    if constexpr (!FULL) {
        // battery + PV + non-shiftable load only: no outage, no thermal end uses.
        float eb = 0.0f;
        if constexpr (F64) {
            if (has_batt) eb = battery_step_f64(B.p, a.es, INFINITY, t == 0, S);
        } else if constexpr (PREC == 2) {
            if (has_batt) { BattC bc; load_battc(bc, B.p); eb = battery_charge_chain(bc, a.es, INFINITY, S); }
        } else if (has_batt) eb = battery_step(B.batt, a.es, INFINITY, S);
        // t = 0: the load is booked at reset, by the step, and again by update_variables
        const float c_ns = first ? 3.0f * R.nsl : R.nsl;
        const float c_b = first ? 2.0f * eb : eb;
        if constexpr (CHECK) { if (!(R.nsl >= 0.0f)) viol |= CLV_NSL; O.viol = viol; }      // update_electricity_consumption (energy_model.py:146-148)
        const float net = fmaf(c_ns + c_b, B.r, R.sol);          // (explicit: the lean kernels restate this line)
        O.net = net; O.cost = mul_rn(net, R.price); O.emission = fmaxf(0.0f, net * R.carbon);
        O.eb = eb; O.cool_dem = 0.0f; O.heat_dem = 0.0f; O.dhw_dem = 0.0f; O.c_cool = 0.0f; O.c_heat = 0.0f; O.c_dhw = 0.0f; O.c_ns = c_ns * B.r;
        O.base_net = net - c_b * B.r; O.net_ws = O.base_net; O.expected = R.nsl; O.served = R.nsl;
        O.se_cool = O.se_heat = O.se_dhw = 0.0f;
        return;
    } else {
        Acc A = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        const bool heat_hp = B.flags & CLF_HEAT_IS_HP;
        const float t0_iheat = heat_hp ? R.icop_h : B.t0_iheat_div;
        if (first) {
            // reset-time update_variables already booked the ideal loads once (citylearn.py:1884 -> building.py:2618-2652)
            A.c_cool = R.cool * R.icop_c; A.c_heat = R.heat * t0_iheat; A.c_dhw = R.dhw * R.icop_d; A.c_ns = R.nsl;
        }
        // partial-load demand of LSTMDynamicsBuilding (building.py:3080-3158); active from step `lookback + 1`
        float cool_dem = R.cool, heat_dem = R.heat;
        if ((B.flags & CLF_DYNAMICS) && (float)t >= B.dyn_warmup) {
            const bool coh = B.a_coh >= 0;
            if (B.a_cd >= 0 || coh) {
                const bool on = R.hvac == 1.0f || R.hvac == 3.0f;
                cool_dem = on ? fminf(a.cd * B.cd_pow * B.dt, B.cd_pow - A.c_cool * B.r) * R.cop_c : 0.0f;
            }
            if (B.a_hd >= 0 || coh) {
                const bool on = R.hvac == 2.0f || R.hvac == 3.0f;
                heat_dem = on ? fminf(a.hd * B.hd_pow, B.hd_pow - A.c_heat * B.r) * R.cop_h : 0.0f;
            }
        }
        float eb_b = 0.0f;
        const bool es_first = a.es < 0.0f;                         // building.py:1606-1609
        if (has_batt && R.outage) {                                // the order only matters through `flexibility`
            if (es_first) {
                if constexpr (F64) eb_b = battery_step_f64(B.p, a.es, flexibility<CHECK>(B, R, A, &viol), t == 0, S);
                else if constexpr (PREC == 2) { BattC bc; load_battc(bc, B.p); eb_b = battery_charge_chain(bc, a.es, flexibility<CHECK>(B, R, A, &viol), S); }
                else { BattP bp; load_batt(bp, B.p); eb_b = battery_step(bp, a.es, flexibility<CHECK>(B, R, A, &viol), S); }
                A.c_b += eb_b;
            }
        }
        float eb_cs = 0.0f, eb_hs = 0.0f, eb_ds = 0.0f, e_cool = cool_dem, e_heat = heat_dem, e_dhw = R.dhw;
        const float ir = rcp(B.r);
        // each end use is skipped when the building has neither the device nor the tank (its demand is then zero in
        // every valid schema: the reference asserts demand <= device output, building.py:1825-1829)
        if (B.flags & (CLF_COOL_DEV | CLF_COOL_STO)) {
            TankP T; load_tank(T, B.p, CLP_CS_CAP, CLP_CS_IRTE);
            end_use<CHECK>(B, R, A, A.c_cool, cool_dem, a.cs, T.cap, B.cd_pow, R.cop_c, R.icop_c, T, ir, S.cs, eb_cs, e_cool, &viol, CLV_COOLING);
        }
        if (B.flags & (CLF_HEAT_DEV | CLF_HEAT_STO)) {
            TankP T; load_tank(T, B.p, CLP_HS_CAP, CLP_HS_IRTE);
            end_use<CHECK>(B, R, A, A.c_heat, heat_dem, a.hs, pw(B.p, CLP_CS_CAP) * B.dt /* sic, building.py:1720 */, B.hd_pow, R.cop_h,
                           R.icop_h, T, ir, S.hs, eb_hs, e_heat, &viol, CLV_HEATING);
        }
        if (B.flags & (CLF_DHW_DEV | CLF_DHW_STO)) {
            TankP T; load_tank(T, B.p, CLP_DS_CAP, CLP_DS_IRTE);
            end_use<CHECK>(B, R, A, A.c_dhw, R.dhw, a.ds, pw(B.p, CLP_HS_CAP) * B.dt /* sic, building.py:1765 */, B.dd_pow, R.cop_d,
                           R.icop_d, T, ir, S.ds, eb_ds, e_dhw, &viol, CLV_DHW);
        }
        // non-shiftable load (building.py:1784-1789)
        const float e_ns = fminf(R.nsl, flexibility<CHECK>(B, R, A, &viol));
        if constexpr (CHECK) { if (!(e_ns >= 0.0f)) viol |= CLV_NSL; }          // update_electricity_consumption's own assertion (energy_model.py:146-148)
        A.c_ns += e_ns;
        if (has_batt && !(R.outage && es_first)) {
            if constexpr (F64) eb_b = battery_step_f64(B.p, a.es, flexibility<CHECK>(B, R, A, &viol), t == 0, S);
            else if constexpr (PREC == 2) { BattC bc; load_battc(bc, B.p); eb_b = battery_charge_chain(bc, a.es, flexibility<CHECK>(B, R, A, &viol), S); }
            else { BattP bp; load_batt(bp, B.p); eb_b = battery_step(bp, a.es, flexibility<CHECK>(B, R, A, &viol), S); }
            A.c_b += eb_b;
        }
        if (first) {
            // the first step's update_variables runs the t == 0 block again (building.py:2618-2652)
            A.c_cool += (e_cool + eb_cs) * R.icop_c;
            A.c_heat += (e_heat + eb_hs) * t0_iheat;
            A.c_dhw += (e_dhw + eb_ds) * R.icop_d;
            A.c_ns += e_ns;
            A.c_b += eb_b;
        }
        const float net = R.outage ? 0.0f : (A.c_cool + A.c_heat + A.c_dhw + A.c_ns + A.c_b) * B.r + R.sol;
        O.net = net; O.cost = mul_rn(net, R.price); O.emission = fmaxf(0.0f, net * R.carbon);
        if constexpr (CHECK) O.viol = viol;
        O.eb = eb_b;
        O.cool_dem = e_cool + fabsf(fminf(eb_cs, 0.0f));          // building.py:1435-1437
        O.heat_dem = e_heat + fabsf(fminf(eb_hs, 0.0f));
        O.dhw_dem = e_dhw + fabsf(fminf(eb_ds, 0.0f));
        // what Device.electricity_consumption reports: accumulator * time_step_ratio (energy_model.py:118)
        O.c_cool = A.c_cool * B.r; O.c_heat = A.c_heat * B.r; O.c_dhw = A.c_dhw * B.r; O.c_ns = A.c_ns * B.r;
        // evaluate()'s baseline: remove what the storages did (building.py:345-366, 413-463) and, for dynamics
        // buildings, add back the ideal-vs-delivered load difference (building.py:2877-2905)
        O.se_cool = eb_cs * R.icop_c; O.se_heat = eb_hs * R.icop_h; O.se_dhw = eb_ds * R.icop_d;      // building.py:413-457
        float base = net - (eb_cs * R.icop_c + eb_hs * R.icop_h + eb_ds * R.icop_d + A.c_b * B.r);
        O.net_ws = base;
        // (the heating difference of every step is converted with ONE COP, that of the step evaluate() is called at: building.py:2893-2898)
        if (B.flags & CLF_DYNAMICS) base += (R.cool - cool_dem) * R.icop_c + (R.heat - heat_dem) * (heat_hp ? R.icop_h_eval : B.t0_iheat_div);
        O.base_net = base;
        O.expected = cool_dem + heat_dem + R.dhw + R.nsl;
        O.served = e_cool + fmaxf(-eb_cs, 0.0f) + e_heat + fmaxf(-eb_hs, 0.0f) + e_dhw + fmaxf(-eb_ds, 0.0f) + e_ns;
    }
}

// Chargers + washing machines of this building (cl_flex.h planes): added to the net after every other device
// (building.py:2657-2693; a power outage zeroes the net, chargers included) and the chargers' part removed again from
// evaluate()'s baseline (building.py:345-366).
CL_DEV void apply_flex(bool outage, float price, float carbon, float load, float chargers, Out& O) {
    if (!outage) {
        O.net += load;
        O.cost = mul_rn(O.net, price);
        O.emission = fmaxf(0.0f, O.net * carbon);
        O.base_net += load;
        O.net_ws += load;
    }
    O.base_net -= chargers;
    O.net_ws -= chargers;
}

// Electric_Vehicles_Reward_Function for one building (reward_function.py:415-531): `marl` is MARL's reward for it,
// K* the net-independent sums cl_flex_kernel left; buildings without chargers get 0.
CL_DEV float ev_reward(bool has_chargers, float marl, float net, float k0, float kneg, float kpos) {
    if (!has_chargers) return 0.0f;
    const float k = k0 + (net < 0.0f ? kneg : (net > 0.0f ? kpos : 0.0f));
    return k / (1.0f + fabsf(marl));
}

// Per-building reward from this unit's own quantities (reward_function.py); MARL needs the district sum
// and is finished by the caller.  `kind` is wave-uniform.
template <bool FULL>
CL_DEV float unit_reward(int kind, const Bp& B, const State& S, float net) {
    switch (kind) {
    case CLR_INDEPENDENT_SAC: return fminf(-net, 0.0f);
    case CLR_SOLAR_PENALTY: {
        const float sg = net > 0.0f ? 1.0f : (net < 0.0f ? -1.0f : 0.0f), an = fabsf(net);
        float rw = pw(B.p, CLP_L_CAP) > CL_ZDP ? -(1.0f + sg * S.soc) * an : 0.0f;
        if constexpr (FULL) {
            rw += pw(B.p, CLP_CS_CAP) > CL_ZDP ? -(1.0f + sg * S.cs) * an : 0.0f;
            rw += pw(B.p, CLP_HS_CAP) > CL_ZDP ? -(1.0f + sg * S.hs) * an : 0.0f;
            rw += pw(B.p, CLP_DS_CAP) > CL_ZDP ? -(1.0f + sg * S.ds) * an : 0.0f;
        }
        return rw;
    }
    case CLR_MARL: case CLR_EV: return net;   // placeholder, finished with the district sum
    default: {
        const float m = fmaxf(net, 0.0f);
        return B.rw_exponent == 1.0f ? -m : -__powf(m, B.rw_exponent);
    }
    }
}

// unit_reward<false> for the N envs of a lane with the (wave-uniform) switch outside the env loop: same expressions, same bits.
template <int N>
CL_DEV void lean_rewards(int kind, const Bp& B, const float (&soc)[N], const float (&net)[N], float (&rw)[N]) {
    switch (kind) {
    case CLR_INDEPENDENT_SAC:
#pragma unroll
        for (int i = 0; i < N; ++i) rw[i] = fminf(-net[i], 0.0f);
        break;
    case CLR_SOLAR_PENALTY: {
        const bool has = pw(B.p, CLP_L_CAP) > CL_ZDP;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float sg = net[i] > 0.0f ? 1.0f : (net[i] < 0.0f ? -1.0f : 0.0f), an = fabsf(net[i]);
            rw[i] = has ? -(1.0f + sg * soc[i]) * an : 0.0f;
        }
        break;
    }
    case CLR_MARL: case CLR_EV:
#pragma unroll
        for (int i = 0; i < N; ++i) rw[i] = net[i];
        break;
    default:
        if (B.rw_exponent == 1.0f) {
#pragma unroll
            for (int i = 0; i < N; ++i) rw[i] = -fmaxf(net[i], 0.0f);
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) rw[i] = -__powf(fmaxf(net[i], 0.0f), B.rw_exponent);
        }
    }
}

CL_DEV float marl_reward(float net, float district_net) {
    // reward_function.py:132-143: sign(-net) * 0.01 * net^2 * max(0, district)
    const float sg = net < 0.0f ? 1.0f : (net > 0.0f ? -1.0f : 0.0f);
    return sg * 0.01f * net * net * fmaxf(0.0f, district_net);
}

}  // namespace cl
