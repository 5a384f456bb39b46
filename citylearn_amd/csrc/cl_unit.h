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
// per-lane work is straight-line select/min/max code -- the only divergent branches are the rare
// power-outage ordering cases.  Divisions by wave-uniform values use host-precomputed reciprocals;
// per-lane 1/x and sqrt use the hardware v_rcp_f32 / v_sqrt_f32 (1 ulp), well inside the 1e-4 parity bar.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/citylearn_amd.h"

#define CL_DEV __device__ __forceinline__
#define CL_ZDP 1e-6f          // data.py:19 ZERO_DIVISION_PLACEHOLDER

namespace cl {

CL_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
CL_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
CL_DEV float sel(bool c, float a, float b) { return c ? a : b; }

// Per-building parameters, wave-uniform.  Raw words come from `params[b][*]` via scalar loads; the derived
// reciprocals / slopes are recomputed per wave (a handful of uniform VALU ops, amortised over the step).
struct Bp {
    uint32_t flags;
    float dt, r;
    // battery
    float cap, pow, loss, clc, dod, eff0, soc0;
    float cpc_x1, cpc_y0, cpc_y1, cpc_s0, cpc_s1;                    // 3-point curve: y0 + s0*x | y1 + s1*(x-x1)
    float pec_x1, pec_x2, pec_x3, pec_y0, pec_y1, pec_y2, pec_y3, pec_s0, pec_s1, pec_s2, pec_s3;
    float inv_cap, inv_pow, one_minus_dod;
    // tanks: capacity, loss, sqrt(eff), 1/sqrt(eff), 1/max(cap,ZDP), max in/out
    float cs_cap, cs_loss, cs_rte, cs_irte, cs_icap, cs_maxin, cs_maxout;
    float hs_cap, hs_loss, hs_rte, hs_irte, hs_icap, hs_maxin, hs_maxout;
    float ds_cap, ds_loss, ds_rte, ds_irte, ds_icap, ds_maxin, ds_maxout;
    float cd_pow, hd_pow, dd_pow, t0_heat_div, dyn_warmup, rw_exponent;
    int a_cs, a_hs, a_ds, a_es, a_cd, a_hd, a_coh;
};

CL_DEV float pw(const uint32_t* __restrict__ p, int slot) { return __uint_as_float(p[slot]); }

CL_DEV void load_bp(Bp& B, const uint32_t* __restrict__ p) {
    B.flags = p[CLP_FLAGS];
    B.dt = pw(p, CLP_DT_HOURS); B.r = pw(p, CLP_TSR);
    B.cap = pw(p, CLP_B_CAP); B.pow = pw(p, CLP_B_POW); B.loss = pw(p, CLP_B_LOSS); B.clc = pw(p, CLP_B_CLC);
    B.dod = pw(p, CLP_B_DOD); B.eff0 = pw(p, CLP_B_EFF0); B.soc0 = pw(p, CLP_B_SOC0);
    const float cx0 = pw(p, CLP_B_CPC_X0), cx1 = pw(p, CLP_B_CPC_X1), cx2 = pw(p, CLP_B_CPC_X2);
    const float cy0 = pw(p, CLP_B_CPC_Y0), cy1 = pw(p, CLP_B_CPC_Y1), cy2 = pw(p, CLP_B_CPC_Y2);
    B.cpc_x1 = cx1; B.cpc_y0 = cy0; B.cpc_y1 = cy1;
    B.cpc_s0 = (cy1 - cy0) / (cx1 - cx0); B.cpc_s1 = (cy2 - cy1) / (cx2 - cx1);
    // NB: the first breakpoint of both curves is 0 in every schema; the interpolation below keeps x0 general
    B.cpc_y0 = cy0 - B.cpc_s0 * cx0;
    const float ex0 = pw(p, CLP_B_PEC_X0), ex1 = pw(p, CLP_B_PEC_X1), ex2 = pw(p, CLP_B_PEC_X2),
                ex3 = pw(p, CLP_B_PEC_X3), ex4 = pw(p, CLP_B_PEC_X4);
    const float ey0 = pw(p, CLP_B_PEC_Y0), ey1 = pw(p, CLP_B_PEC_Y1), ey2 = pw(p, CLP_B_PEC_Y2),
                ey3 = pw(p, CLP_B_PEC_Y3), ey4 = pw(p, CLP_B_PEC_Y4);
    B.pec_x1 = ex1; B.pec_x2 = ex2; B.pec_x3 = ex3;
    B.pec_s0 = (ey1 - ey0) / (ex1 - ex0); B.pec_s1 = (ey2 - ey1) / (ex2 - ex1);
    B.pec_s2 = (ey3 - ey2) / (ex3 - ex2); B.pec_s3 = (ey4 - ey3) / (ex4 - ex3);
    B.pec_y0 = ey0 - B.pec_s0 * ex0; B.pec_y1 = ey1; B.pec_y2 = ey2; B.pec_y3 = ey3;
    B.inv_cap = 1.0f / fmaxf(B.cap, CL_ZDP); B.inv_pow = 1.0f / fmaxf(B.pow, CL_ZDP);
    B.one_minus_dod = 1.0f - B.dod;
#define CL_TANK(px, base)                                                              \
    B.px##_cap = pw(p, base); B.px##_loss = pw(p, base + 1); B.px##_rte = pw(p, base + 2);   \
    B.px##_irte = 1.0f / B.px##_rte; B.px##_icap = 1.0f / fmaxf(B.px##_cap, CL_ZDP);     \
    B.px##_maxin = pw(p, base + 4); B.px##_maxout = pw(p, base + 5);
    CL_TANK(cs, CLP_CS_CAP) CL_TANK(hs, CLP_HS_CAP) CL_TANK(ds, CLP_DS_CAP)
#undef CL_TANK
    B.cd_pow = pw(p, CLP_CD_POW); B.hd_pow = pw(p, CLP_HD_POW); B.dd_pow = pw(p, CLP_DD_POW);
    B.t0_heat_div = pw(p, CLP_T0_HEAT_DIV); B.dyn_warmup = pw(p, CLP_DYN_WARMUP);
    B.rw_exponent = pw(p, CLP_RW_EXPONENT);
    B.a_cs = (int)p[CLP_ACT_COOL_STO]; B.a_hs = (int)p[CLP_ACT_HEAT_STO]; B.a_ds = (int)p[CLP_ACT_DHW_STO];
    B.a_es = (int)p[CLP_ACT_ELEC_STO]; B.a_cd = (int)p[CLP_ACT_COOL_DEV]; B.a_hd = (int)p[CLP_ACT_HEAT_DEV];
    B.a_coh = (int)p[CLP_ACT_COH_DEV];
}

// Time-series row of (t, building): wave-uniform.
struct Row {
    float nsl, sol, cool, heat, dhw, cop_c, cop_h, cop_d, icop_c, icop_h, icop_d, price, carbon, hvac;
    bool outage;
};

CL_DEV void load_row(Row& R, const float* __restrict__ q, uint32_t flags) {
    R.nsl = q[CLT_NSL]; R.sol = q[CLT_SOLAR]; R.cool = q[CLT_COOL_DEM]; R.heat = q[CLT_HEAT_DEM];
    R.dhw = q[CLT_DHW_DEM]; R.cop_c = q[CLT_COP_COOL]; R.cop_h = q[CLT_COP_HEAT]; R.cop_d = q[CLT_COP_DHW];
    R.icop_c = 1.0f / R.cop_c; R.icop_h = 1.0f / R.cop_h; R.icop_d = 1.0f / R.cop_d;
    R.price = q[CLT_PRICE]; R.carbon = q[CLT_CARBON]; R.hvac = q[CLT_HVAC_MODE];
    R.outage = (flags & CLF_OUTAGE) && (q[CLT_OUTAGE] != 0.0f);
}

// Carried per-unit state (one lane).
struct State { float soc, eff, degcap, cs, hs, ds; };
// Actions of one unit (inactive -> 0 for storages / NaN-ignored for devices, building.py:1557-1564).
struct Act { float cs, hs, ds, es, cd, hd; };
// Per-unit results of the step.
struct Out { float net, cost, emission, reward, eb, cool_dem, c_cool, c_heat, c_dhw, c_ns; };

// StorageDevice.charge on top of StorageTank.charge's power clamps (energy_model.py:719-768, 850-870).
// `e` is the energy handed to tank.charge() *before* its two time_step_ratio multiplications.
CL_DEV void tank_charge(float e, float prev_soc, float cap, float loss, float rte, float irte, float icap,
                        float maxin, float maxout, float r, float& soc, float& eb) {
    e *= r;
    e = e >= 0.0f ? fminf(e, maxin) : fmaxf(-maxout, e);
    e *= r;
    const float e_init = fmaxf(0.0f, prev_soc * cap * (1.0f - loss));
    const float e_fin = e >= 0.0f ? fminf(e_init + e * rte, cap) : fmaxf(0.0f, e_init + e * irte);
    soc = e_fin * icap;
    const float d = e_fin - e_init;
    eb = d >= 0.0f ? d * irte : d * rte;
}

struct Acc { float c_cool, c_heat, c_dhw, c_ns, c_b; };

CL_DEV float flexibility(const Bp& B, const Row& R, const Acc& A) {
    // building.py:640-668 (only evaluated under outage; +inf otherwise)
    if (!R.outage) return INFINITY;
    const float used = (A.c_cool + A.c_heat + A.c_dhw + A.c_ns + A.c_b) * B.r;
    return fmaxf(0.0f, fabsf(R.sol) - used);
}

// Battery.charge (energy_model.py:1027-1057) + update_electrical_storage (building.py:1791-1812).
CL_DEV void battery_step(const Bp& B, const Row& R, float a_es, State& S, Acc& A, float& soc_out, float& eb_out) {
    float E = a_es * B.pow * B.dt;
    E = fminf(E, flexibility(B, R, A));
    const float prev = S.soc;
    const float e_init = fmaxf(0.0f, prev * B.cap * (1.0f - B.loss));
    const float socn = e_init * B.inv_cap;
    // capacity_power_curve, `idx = max(0, argmax(x <= xs) - 1)` (energy_model.py:1083-1088)
    const float pmax = B.pow * (socn <= B.cpc_x1 ? B.cpc_y0 + B.cpc_s0 * socn : B.cpc_y1 + B.cpc_s1 * (socn - B.cpc_x1));
    float e;
    if (E >= 0.0f) {
        e = fminf(fminf(pmax, B.pow - A.c_b * B.r), fminf(S.degcap - e_init, E));
    } else {
        const float lim = -fmaxf((prev - B.one_minus_dod) * B.cap * fsqrt(S.eff), 0.0f);
        e = fmaxf(fmaxf(-pmax, lim), E);
    }
    // power_efficiency_curve at min(|E|, pmax)/nominal_power (energy_model.py:1039, 1052, 1103-1107)
    const float x = fabsf(fminf(fabsf(E), pmax)) * B.inv_pow;
    float eff = x <= B.pec_x1 ? B.pec_y0 + B.pec_s0 * x
              : x <= B.pec_x2 ? B.pec_y1 + B.pec_s1 * (x - B.pec_x1)
              : x <= B.pec_x3 ? B.pec_y2 + B.pec_s2 * (x - B.pec_x2)
                              : B.pec_y3 + B.pec_s3 * (x - B.pec_x3);
    const float rte = fsqrt(eff), irte = rcp(rte);
    e *= B.r;
    const float e_fin = e >= 0.0f ? fminf(e_init + e * rte, B.cap) : fmaxf(0.0f, e_init + e * irte);
    const float d = e_fin - e_init;
    const float eb = d >= 0.0f ? d * irte : d * rte;
    // degrade (energy_model.py:1130-1141) with the pre-step degraded capacity
    const float deg = B.clc * B.cap * fabsf(eb) * 0.5f * rcp(fmaxf(S.degcap, CL_ZDP)) * B.r;
    S.degcap = fmaxf(S.degcap - deg, 0.0f);
    S.eff = eff;
    soc_out = e_fin * B.inv_cap;
    eb_out = eb;
    A.c_b += eb;
}

// One end use (cooling / heating / dhw): device + storage in the order given by the storage action's sign.
CL_DEV void end_use(const Bp& B, const Row& R, Acc& A, float& c, float demand, float a_sto, float cscale,
                    float dev_pow, float cop, float icop, float prev_soc, float cap, float loss, float rte,
                    float irte, float icap, float maxin, float maxout, float& soc, float& eb, float& e_dev) {
    const float energy = a_sto * cscale;
    const bool disc = a_sto < 0.0f;                         // storage first (building.py:1611-1622)
    // storage-first lanes discharge now; the others see an untouched tank (energy_balance[t] == 0)
    float soc_a, eb_a;
    tank_charge(fmaxf(-demand, energy) * rcp(B.r), prev_soc, cap, loss, rte, irte, icap, maxin, maxout, B.r, soc_a, eb_a);
    eb_a = disc ? eb_a : 0.0f;
    // device (building.py:1641-1661): c is this end use's accumulator inside A
    float flex = flexibility(B, R, A);
    float max_out = fminf(flex, dev_pow - c * B.r) * cop;
    const float out = fminf(demand - fmaxf(-eb_a, 0.0f), max_out);
    e_dev = out;
    c += fmaxf(0.0f, out * icop);
    // storage after the device (charging or idle lanes; building.py:1663-1687)
    flex = flexibility(B, R, A);
    max_out = fminf(flex, dev_pow - c * B.r) * cop;
    const float e_c = energy > 0.0f ? fminf(max_out, energy) : fmaxf(-demand, energy);
    float soc_c, eb_c;
    tank_charge(e_c * rcp(B.r), prev_soc, cap, loss, rte, irte, icap, maxin, maxout, B.r, soc_c, eb_c);
    soc = disc ? soc_a : soc_c;
    eb = disc ? eb_a : eb_c;
    c += fmaxf(eb, 0.0f) * icop;
}

// The whole unit step.  `t` and `t0_quirk` are wave-uniform.
CL_DEV void unit_step(const Bp& B, const Row& R, int t, bool t0_quirk, const Act& a, State& S, Out& O) {
    Acc A = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const bool first = t0_quirk && t == 0;
    const bool heat_hp = B.flags & CLF_HEAT_IS_HP;
    const float t0_iheat = heat_hp ? R.icop_h : 1.0f / B.t0_heat_div;
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
    float soc_b = S.soc, eb_b = 0.0f;
    const bool has_batt = B.flags & CLF_BATTERY;
    const bool es_first = a.es < 0.0f;                         // building.py:1606-1609
    if (has_batt && R.outage) {                                // order only matters through `flexibility`
        if (es_first) battery_step(B, R, a.es, S, A, soc_b, eb_b);
    }
    float eb_cs = 0.0f, eb_hs = 0.0f, eb_ds = 0.0f, e_cool = cool_dem, e_heat = heat_dem, e_dhw = R.dhw;
    if (B.flags & CLF_THERMAL) {
        end_use(B, R, A, A.c_cool, cool_dem, a.cs, B.cs_cap, B.cd_pow, R.cop_c, R.icop_c, S.cs, B.cs_cap, B.cs_loss,
                B.cs_rte, B.cs_irte, B.cs_icap, B.cs_maxin, B.cs_maxout, S.cs, eb_cs, e_cool);
        end_use(B, R, A, A.c_heat, heat_dem, a.hs, B.cs_cap * B.dt /* sic, building.py:1720 */, B.hd_pow, R.cop_h,
                R.icop_h, S.hs, B.hs_cap, B.hs_loss, B.hs_rte, B.hs_irte, B.hs_icap, B.hs_maxin, B.hs_maxout, S.hs,
                eb_hs, e_heat);
        end_use(B, R, A, A.c_dhw, R.dhw, a.ds, B.hs_cap * B.dt /* sic, building.py:1765 */, B.dd_pow, R.cop_d,
                R.icop_d, S.ds, B.ds_cap, B.ds_loss, B.ds_rte, B.ds_irte, B.ds_icap, B.ds_maxin, B.ds_maxout, S.ds,
                eb_ds, e_dhw);
    }
    // non-shiftable load (building.py:1784-1789)
    const float e_ns = fminf(R.nsl, flexibility(B, R, A));
    A.c_ns += e_ns;
    if (has_batt && !(R.outage && es_first)) battery_step(B, R, a.es, S, A, soc_b, eb_b);
    S.soc = soc_b;
    if (first) {
        // the first step's update_variables runs the t == 0 block again (building.py:2618-2652)
        A.c_cool += (e_cool + eb_cs) * R.icop_c;
        A.c_heat += (e_heat + eb_hs) * t0_iheat;
        A.c_dhw += (e_dhw + eb_ds) * R.icop_d;
        A.c_ns += e_ns;
        A.c_b += eb_b;
    }
    const float net = R.outage ? 0.0f : (A.c_cool + A.c_heat + A.c_dhw + A.c_ns + A.c_b) * B.r + R.sol;
    O.net = net;
    O.cost = net * R.price;
    O.emission = fmaxf(0.0f, net * R.carbon);
    O.eb = eb_b;
    O.cool_dem = e_cool + fabsf(fminf(eb_cs, 0.0f));          // building.py:1435
    O.c_cool = A.c_cool; O.c_heat = A.c_heat; O.c_dhw = A.c_dhw; O.c_ns = A.c_ns;
}

// Per-building reward from this unit's own quantities (reward_function.py); MARL needs the district sum
// and is finished by the caller.
CL_DEV float unit_reward(int kind, const Bp& B, const State& S, float net) {
    switch (kind) {
    case CLR_INDEPENDENT_SAC: return fminf(-net, 0.0f);
    case CLR_SOLAR_PENALTY: {
        const float sg = net > 0.0f ? 1.0f : (net < 0.0f ? -1.0f : 0.0f), an = fabsf(net);
        float rw = 0.0f;
        rw += B.cs_cap > CL_ZDP ? -(1.0f + sg * S.cs) * an : 0.0f;
        rw += B.hs_cap > CL_ZDP ? -(1.0f + sg * S.hs) * an : 0.0f;
        rw += B.ds_cap > CL_ZDP ? -(1.0f + sg * S.ds) * an : 0.0f;
        rw += B.cap > CL_ZDP ? -(1.0f + sg * S.soc) * an : 0.0f;
        return rw;
    }
    case CLR_MARL: return net;   // placeholder, finished with the district sum
    default: {
        const float m = fmaxf(net, 0.0f);
        return B.rw_exponent == 1.0f ? -m : -__powf(m, B.rw_exponent);
    }
    }
}

CL_DEV float marl_reward(float net, float district_net) {
    // reward_function.py:132-143: sign(-net) * 0.01 * net^2 * max(0, district)
    const float sg = net < 0.0f ? 1.0f : (net > 0.0f ? -1.0f : 0.0f);
    return sg * 0.01f * net * net * fmaxf(0.0f, district_net);
}

}  // namespace cl
