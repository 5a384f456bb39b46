// cl_full.h -- thermal / outage / partial-load districts (heat pump, heater, three tanks, battery, power outage, LSTM
// partial-load demand): the unit arithmetic written once over a "pack" type F (one env per lane: float; two envs per lane:
// float2) + the step kernel built on it.  Included by cl_kernels.hip after district_reduce.
//
// Same reference semantics as cl::unit_step<true> in cl_unit.h (building.py:1500-1634, 1641-1812, 640-668, 3080-3158;
// energy_model.py:719-768, 850-870, 1027-1141; building.py:2615-2703) -- what changed is the instruction count.  PMC of the
// round-1 kernel (profiles/r02_thermal_*): 384 VALU instructions per (env, building) unit, the vector ALUs busy 53 % of an
// 11 us launch whose waves all load, then all compute, then all store -- the launch was VALU-bound between two memory
// round trips, not HBM-bound.  Three things cut the count, none changes a lane's arithmetic (every fused multiply-add
// is explicit and contraction is off, so VEC = 1 and VEC = 2 give the same bits):
//   * OUT (power outage on this (t, building) row) is wave-uniform: the common no-outage instantiation carries none of the
//     `downward_electrical_flexibility` bookkeeping (eight evaluations of max(0, |solar| - sum of five consumptions));
//   * one StorageTank.charge per end use instead of two: the discharge-first lanes and the charge-after-device lanes hand the
//     tank the same expression  e = energy > 0 ? min(device headroom, energy) : max(-demand, energy);  only the charging
//     clamp needs the device's consumption first, and that costs four instructions, not a second tank update;
//   * two envs per lane: the wave-uniform work (parameter moves, address arithmetic, uniform compares) is shared by both envs
//     and mul / add / fma run as v_pk_*_f32.
#pragma once

#pragma clang fp contract(off)

namespace clv {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

template <typename F> struct Tr;
template <> struct Tr<float> { using M = bool; static constexpr int N = 1; };
template <> struct Tr<f2> { using M = i2; static constexpr int N = 2; };

CL_DEV float vmin(float a, float b) { return fminf(a, b); }
CL_DEV f2 vmin(f2 a, f2 b) { return __builtin_elementwise_min(a, b); }
CL_DEV float vmax(float a, float b) { return fmaxf(a, b); }
CL_DEV f2 vmax(f2 a, f2 b) { return __builtin_elementwise_max(a, b); }
CL_DEV float vabs(float a) { return fabsf(a); }
CL_DEV f2 vabs(f2 a) { return __builtin_elementwise_abs(a); }
CL_DEV float vfma(float a, float b, float c) { return fmaf(a, b, c); }
CL_DEV f2 vfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
CL_DEV float vsel(bool m, float a, float b) { return m ? a : b; }
CL_DEV f2 vsel(i2 m, f2 a, f2 b) { return m ? a : b; }
CL_DEV float vrcp(float a) { return cl::rcp(a); }
CL_DEV f2 vrcp(f2 a) { f2 r; r.x = cl::rcp(a.x); r.y = cl::rcp(a.y); return r; }
CL_DEV float vrsq(float a) { return cl::rsq(a); }
CL_DEV f2 vrsq(f2 a) { f2 r; r.x = cl::rsq(a.x); r.y = cl::rsq(a.y); return r; }
CL_DEV float vsqrt(float a) { return cl::fsqrt(a); }
CL_DEV f2 vsqrt(f2 a) { f2 r; r.x = cl::fsqrt(a.x); r.y = cl::fsqrt(a.y); return r; }
CL_DEV float vmed3(float a, float b, float c) { return cl::med3(a, b, c); }
CL_DEV f2 vmed3(f2 a, f2 b, f2 c) { f2 r; r.x = cl::med3(a.x, b.x, c.x); r.y = cl::med3(a.y, b.y, c.y); return r; }
CL_DEV float vpow(float a, float e) { return __powf(a, e); }
CL_DEV f2 vpow(f2 a, float e) { f2 r; r.x = __powf(a.x, e); r.y = __powf(a.y, e); return r; }
template <typename F> CL_DEV F splat(float s) { return (F)(s); }

// Wave-uniform parameters of one building, from its compact CLP_F_* block (include/citylearn_amd.h).  The head (8 words) travels
// with the row; each end use fetches its tank (8 words, one scalar load) and the battery its 24 words where they are consumed --
// while other waves of the SIMD compute -- so that no more than one group occupies SGPRs at a time.  (Fetching head + three tanks
// + row as ONE batch was tried: 54 SGPRs at once on top of the kernel arguments; the allocator parked 16 of them in VGPR lanes
// and the v_writelane / v_readlane traffic added ~100 VALU instructions per building.)
struct FP {
    const uint32_t* __restrict__ f;
    uint32_t flags;
    int a_cd, a_hd, a_coh;
    float dt, r, cd_pow, hd_pow, dd_pow, t0_iheat_div, dyn_warmup, rw_exponent;
};

// LP: `f` is the workgroup's LDS copy of the block (cl_step_full_kernel<.., LP = true>).  The words a wave branches on or forms
// addresses from go back to SGPRs; the float parameters stay in VGPRs, where a VALU instruction can read any number of them.
template <bool LP>
CL_DEV uint32_t uword(const uint32_t* __restrict__ f, int k) {
    if constexpr (LP) return (uint32_t)__builtin_amdgcn_readfirstlane((int)f[k]);
    else return f[k];
}

// (Tried instead of LP for the launches that are not building-chunked: the tank and battery words fetched by VECTOR loads from the
//  uniform block address -- no staging barrier, no v_mov per two-scalar select, SGPR spill traffic 80 -> 8 lane moves, 11 % fewer
//  VALU instructions in the loop -- 2020 schema 9 x 65 536: 8.62 vs 8.65 us.  That launch is not bound by instruction issue but by
//  its phases: scripts/wave_timeline.py shows the 21 MB read burst of the first buildings taking 2.4 us to reach the last wave.)
template <bool LP = false>
CL_DEV void load_fp(FP& P, const uint32_t* __restrict__ f) {
    P.f = f;
    P.flags = uword<LP>(f, 0); P.a_cd = (int)uword<LP>(f, 5); P.a_hd = (int)uword<LP>(f, 6); P.a_coh = (int)uword<LP>(f, 7);
    P.dt = cl::pw(f, 8); P.r = cl::pw(f, 9); P.cd_pow = cl::pw(f, 10); P.hd_pow = cl::pw(f, 11); P.dd_pow = cl::pw(f, 12);
    P.t0_iheat_div = cl::pw(f, 13); P.dyn_warmup = cl::pw(f, 14); P.rw_exponent = cl::pw(f, 15);
}

// tank k = 0 cooling / 1 heating / 2 dhw; `cscale`: kWh per unit storage action
CL_DEV void load_tank_f(cl::TankP& T, float& cscale, const uint32_t* __restrict__ f, int k) {
    const int o = (CLP_F_TANK - CLP_F_FIRST) + 8 * k;
    T.cap = cl::pw(f, o); T.capl = cl::pw(f, o + 1); T.rte = cl::pw(f, o + 2); T.irte = cl::pw(f, o + 3);
    T.icap = cl::pw(f, o + 4); T.maxin = cl::pw(f, o + 5); T.maxout = cl::pw(f, o + 6); cscale = cl::pw(f, o + 7);
}

CL_DEV void load_batt_f(cl::BattP& B, const uint32_t* __restrict__ q, float r) {
    B.r = r; B.pdt = cl::pw(q, 0); B.pow = cl::pw(q, 1); B.cap = cl::pw(q, 2); B.capl = cl::pw(q, 3); B.inv_cap = cl::pw(q, 4);
    B.inv_pow = cl::pw(q, 5); B.omd = cl::pw(q, 6); B.degk = cl::pw(q, 7);
    B.cpc_x1 = cl::pw(q, 8); B.cpc_a0 = cl::pw(q, 9); B.cpc_b0 = cl::pw(q, 10); B.cpc_a1 = cl::pw(q, 11); B.cpc_b1 = cl::pw(q, 12);
    B.pec_x1 = cl::pw(q, 13); B.pec_x2 = cl::pw(q, 14); B.pec_x3 = cl::pw(q, 15);
    B.pec_a0 = cl::pw(q, 16); B.pec_b0 = cl::pw(q, 17); B.pec_a1 = cl::pw(q, 18); B.pec_b1 = cl::pw(q, 19);
    B.pec_a2 = cl::pw(q, 20); B.pec_b2 = cl::pw(q, 21); B.pec_a3 = cl::pw(q, 22); B.pec_b3 = cl::pw(q, 23);
}

template <typename F> struct St { F soc, eff, degcap, cs, hs, ds; };                     // carried state of the lane's env(s)
template <typename F> struct Ac { F cs, hs, ds, es, cd, hd; };                            // actions (inactive -> 0)
template <typename F> struct Ou { F net, cost, emission, eb, cool_dem, heat_dem, dhw_dem, c_cool, c_heat, c_dhw, c_ns, base_net, expected, served, net_ws, se_cool, se_heat, se_dhw; };
template <typename F> struct Ax { F c_cool, c_heat, c_dhw, c_ns, c_b; };                  // running electricity_consumption[t]

// building.py:640-668 during an outage: max(0, |solar| - consumption so far)
template <typename F>
CL_DEV F flexibility(const FP& B, const cl::Row& R, const Ax<F>& A) {
    const F sum = A.c_cool + A.c_heat + A.c_dhw + A.c_ns + A.c_b;
    return vmax(splat<F>(0.0f), vfma(-sum, splat<F>(B.r), splat<F>(fabsf(R.sol))));
}

// Battery.charge(E) (energy_model.py:1027-1141); E already carries Battery.charge's own `* time_step_ratio`.
template <typename F>
CL_DEV F battery_energy(const cl::BattP& B, F E, F& soc, F& eff_s, F& degcap) {
    const F zero = splat<F>(0.0f);
    const F prev = soc;
    const F e_init = vmax(zero, prev * B.capl);
    const F socn = e_init * B.inv_cap;
    const auto lo = socn <= B.cpc_x1;
    const F pmax = vfma(vsel(lo, splat<F>(B.cpc_b0), splat<F>(B.cpc_b1)), socn, vsel(lo, splat<F>(B.cpc_a0), splat<F>(B.cpc_a1)));
    const F e_chg = vmin(vmin(pmax, splat<F>(B.pow)), vmin(degcap - e_init, E));
    const F lim = -vmax((prev - B.omd) * B.cap * vsqrt(eff_s), zero);                    // previous call's efficiency
    const F e_dis = vmax(vmax(-pmax, lim), E);
    F e = vsel(E >= 0.0f, e_chg, e_dis);
    const F x = vabs(vmin(vabs(E), pmax)) * B.inv_pow;
    const auto s0 = x <= B.pec_x1, s1 = x <= B.pec_x2, s2 = x <= B.pec_x3;
    const F eb_ = vsel(s0, splat<F>(B.pec_b0), vsel(s1, splat<F>(B.pec_b1), vsel(s2, splat<F>(B.pec_b2), splat<F>(B.pec_b3))));
    const F ea_ = vsel(s0, splat<F>(B.pec_a0), vsel(s1, splat<F>(B.pec_a1), vsel(s2, splat<F>(B.pec_a2), splat<F>(B.pec_a3))));
    const F eff = vfma(eb_, x, ea_);
    const F irte = vrsq(eff), rte = eff * irte;
    e = e * B.r;
    const auto chg = e >= 0.0f;
    const F e_fin = vmed3(vfma(e, vsel(chg, rte, irte), e_init), zero, splat<F>(B.cap));
    const F d = e_fin - e_init;
    const F eb = d * vsel(chg, irte, rte);
    degcap = vmax(vfma(-(B.degk * vabs(eb)), vrcp(vmax(degcap, splat<F>(CL_ZDP))), degcap), zero);
    eff_s = eff;
    soc = e_fin * B.inv_cap;
    return eb;
}

// update_electrical_storage (building.py:1801-1812) under CLD_F64_CHAIN: cl::battery_charge_chain for the lane's env (`deg` is then the capacity
// loss).  `flex`: the downward flexibility where OUT, else unused.  `g`: the building's whole parameter row in global memory (the CLP_C_* block).
// One env per lane only: at two envs per lane every instantiation parks 24 - 176 bytes per lane in scratch -- the multi-tile kernel 16 bytes of its
// table-row struct (the two-envs-per-lane kernels' old fragility), the chunked one plain register pressure under its 128-register cap -- however the
// pack is taken apart; the host launches the chain's thermal kernels at one env per lane.
template <typename F, bool OUT>
CL_DEV F battery_chain(const uint32_t* __restrict__ g, F a_es, F flex, F& soc, F& eff, F& deg) {
    static_assert(Tr<F>::N == 1, "CLD_F64_CHAIN in the pack-generic unit: one env per lane");
    cl::BattC bc;
    cl::load_battc(bc, g);
    cl::State S = {soc, eff, deg, 0.0f, 0.0f, 0.0f};
    const F eb = cl::battery_charge_chain(bc, a_es, OUT ? flex : INFINITY, S);
    soc = S.soc; eff = S.eff; deg = S.degcap;
    return eb;
}

// StorageDevice.charge under StorageTank.charge's power clamps (energy_model.py:719-768, 850-870)
template <typename F>
CL_DEV void tank_charge(F e, F prev_soc, const cl::TankP& T, float r, F& soc, F& eb) {
    const F zero = splat<F>(0.0f);
    e = e * r;
    e = vmed3(e, splat<F>(-T.maxout), splat<F>(T.maxin));                 // e >= 0 ? min(e, maxin) : max(-maxout, e)
    e = e * r;
    const F e_init = vmax(zero, prev_soc * T.capl);
    // charge: min(e_init + e rte, cap); discharge: max(0, e_init + e / rte).  0 <= e_init <= cap, so the unused bound of either
    // branch is inactive and both are one clamp of one fma (the same collapse as in cl::battery_energy)
    const F e_fin = vmed3(vfma(e, vsel(e >= 0.0f, splat<F>(T.rte), splat<F>(T.irte)), e_init), zero, splat<F>(T.cap));
    soc = e_fin * T.icap;
    const F d = e_fin - e_init;
    eb = d * vsel(d >= 0.0f, splat<F>(T.irte), splat<F>(T.rte));
}

// One end use: device + tank in the order the storage action's sign gives (building.py:1611-1622, 1641-1687).
// `demand` may be per lane (partial-load demand) or a splat of the row value.
template <typename F, bool OUT>
CL_DEV void end_use(const FP& B, const cl::Row& R, Ax<F>& A, F& c, F demand, F a_sto, float cscale, float dev_pow, float cop,
                    float icop, const cl::TankP& T, float ir, F& soc, F& eb, F& e_dev) {
    const F zero = splat<F>(0.0f);
    const F energy = a_sto * cscale;
    // device headroom before anything of this end use ran (a discharge-first tank books nothing until the end)
    F lim = vfma(-c, splat<F>(B.r), splat<F>(dev_pow));
    if constexpr (OUT) lim = vmin(flexibility<F>(B, R, A), lim);
    const F max_out = lim * cop;
    // ... and after the device served the whole demand (what a charging tank may still draw through it)
    const F c_keep = c;
    c = c + vmax(zero, vmin(demand, max_out) * icop);
    F lim2 = vfma(-c, splat<F>(B.r), splat<F>(dev_pow));
    if constexpr (OUT) lim2 = vmin(flexibility<F>(B, R, A), lim2);
    c = c_keep;
    const F e_in = vsel(energy > 0.0f, vmin(lim2 * cop, energy), vmax(-demand, energy));
    tank_charge<F>(e_in * ir, soc, T, B.r, soc, eb);
    // the device covers what a discharging tank did not (building.py:1641-1661)
    const F eb_a = vsel(a_sto < 0.0f, eb, zero);
    const F out = vmin(demand - vmax(-eb_a, zero), max_out);
    e_dev = out;
    c = c + vmax(zero, out * icop);
    c = vfma(vmax(eb, zero), splat<F>(icop), c);
}

// The whole unit step; `t`, `first` (t == 0 under CLD_REF_T0_QUIRK) and OUT are wave-uniform.
template <typename F, bool OUT, bool DETAIL, int PREC = 0>
CL_DEV void unit_step(const FP& B, const cl::Row& R, int t, bool first, const Ac<F>& a, St<F>& S, Ou<F>& O, [[maybe_unused]] const uint32_t* __restrict__ g = nullptr) {
    const F zero = splat<F>(0.0f);
    const bool has_batt = B.flags & CLF_BATTERY;
    Ax<F> A = {zero, zero, zero, zero, zero};
    const bool heat_hp = B.flags & CLF_HEAT_IS_HP;
    const float t0_iheat = heat_hp ? R.icop_h : B.t0_iheat_div;
    if (first) {
        // reset-time update_variables already booked the ideal loads once (citylearn.py:1884 -> building.py:2618-2652)
        A.c_cool = splat<F>(R.cool * R.icop_c); A.c_heat = splat<F>(R.heat * t0_iheat); A.c_dhw = splat<F>(R.dhw * R.icop_d);
        A.c_ns = splat<F>(R.nsl);
    }
    // partial-load demand of LSTMDynamicsBuilding (building.py:3080-3158); active from step `lookback + 1`
    F cool_dem = splat<F>(R.cool), heat_dem = splat<F>(R.heat);
    if ((B.flags & CLF_DYNAMICS) && (float)t >= B.dyn_warmup) {
        const bool coh = B.a_coh >= 0;
        if (B.a_cd >= 0 || coh) {
            const bool on = R.hvac == 1.0f || R.hvac == 3.0f;
            cool_dem = on ? vmin(a.cd * B.cd_pow * B.dt, vfma(-A.c_cool, splat<F>(B.r), splat<F>(B.cd_pow))) * R.cop_c : zero;
        }
        if (B.a_hd >= 0 || coh) {
            const bool on = R.hvac == 2.0f || R.hvac == 3.0f;
            heat_dem = on ? vmin(a.hd * B.hd_pow, vfma(-A.c_heat, splat<F>(B.r), splat<F>(B.hd_pow))) * R.cop_h : zero;
        }
    }
    // battery first where its action is negative (building.py:1606-1609): the order only matters through the outage coupling
    F eb_first = zero;
    St<F> S_first = S;
    const auto es_first = a.es < 0.0f;
    if constexpr (OUT) {
        if (has_batt) {
            if constexpr (PREC == 2) eb_first = battery_chain<F, true>(g, a.es, flexibility<F>(B, R, A), S_first.soc, S_first.eff, S_first.degcap);
            else {
                cl::BattP bp; load_batt_f(bp, B.f + (CLP_F_BATT - CLP_F_FIRST), B.r);
                eb_first = battery_energy<F>(bp, vmin(a.es * bp.pdt, flexibility<F>(B, R, A)), S_first.soc, S_first.eff, S_first.degcap);
            }
            A.c_b = A.c_b + vsel(es_first, eb_first, zero);
        }
    }
    F eb_cs = zero, eb_hs = zero, eb_ds = zero, e_cool = cool_dem, e_heat = heat_dem, e_dhw = splat<F>(R.dhw);
    const float ir = cl::rcp(B.r);
    // action scales: cooling by its own capacity, heating by the COOLING capacity, dhw by the HEATING capacity (sic, building.py:1676, 1720, 1765)
    if (B.flags & (CLF_COOL_DEV | CLF_COOL_STO)) {
        cl::TankP T; float sc; load_tank_f(T, sc, B.f, 0);
        end_use<F, OUT>(B, R, A, A.c_cool, cool_dem, a.cs, sc, B.cd_pow, R.cop_c, R.icop_c, T, ir, S.cs, eb_cs, e_cool);
    }
    if (B.flags & (CLF_HEAT_DEV | CLF_HEAT_STO)) {
        cl::TankP T; float sc; load_tank_f(T, sc, B.f, 1);
        end_use<F, OUT>(B, R, A, A.c_heat, heat_dem, a.hs, sc, B.hd_pow, R.cop_h, R.icop_h, T, ir, S.hs, eb_hs, e_heat);
    }
    if (B.flags & (CLF_DHW_DEV | CLF_DHW_STO)) {
        cl::TankP T; float sc; load_tank_f(T, sc, B.f, 2);
        end_use<F, OUT>(B, R, A, A.c_dhw, splat<F>(R.dhw), a.ds, sc, B.dd_pow, R.cop_d, R.icop_d, T, ir, S.ds, eb_ds, e_dhw);
    }
    // non-shiftable load (building.py:1784-1789)
    F e_ns = splat<F>(R.nsl);
    if constexpr (OUT) e_ns = vmin(e_ns, flexibility<F>(B, R, A));
    A.c_ns = A.c_ns + e_ns;
    F eb_b = zero;
    if (has_batt) {
        F soc, eff, deg, eb_last;
        if constexpr (PREC == 2) {
            soc = S.soc; eff = S.eff; deg = S.degcap;
            F fl = zero;
            if constexpr (OUT) fl = flexibility<F>(B, R, A);
            eb_last = battery_chain<F, OUT>(g, a.es, fl, soc, eff, deg);
        } else {
            cl::BattP bp; load_batt_f(bp, B.f + (CLP_F_BATT - CLP_F_FIRST), B.r);
            F E = a.es * bp.pdt;
            if constexpr (OUT) E = vmin(E, flexibility<F>(B, R, A));
            soc = S.soc; eff = S.eff; deg = S.degcap;
            eb_last = battery_energy<F>(bp, E, soc, eff, deg);
        }
        if constexpr (OUT) {
            eb_b = vsel(es_first, eb_first, eb_last);
            S.soc = vsel(es_first, S_first.soc, soc); S.eff = vsel(es_first, S_first.eff, eff); S.degcap = vsel(es_first, S_first.degcap, deg);
            A.c_b = A.c_b + vsel(es_first, zero, eb_last);
        } else {
            eb_b = eb_last; S.soc = soc; S.eff = eff; S.degcap = deg;
            A.c_b = A.c_b + eb_last;
        }
    }
    if (first) {
        // the first step's update_variables runs the t == 0 block again (building.py:2618-2652)
        A.c_cool = vfma(e_cool + eb_cs, splat<F>(R.icop_c), A.c_cool);
        A.c_heat = vfma(e_heat + eb_hs, splat<F>(t0_iheat), A.c_heat);
        A.c_dhw = vfma(e_dhw + eb_ds, splat<F>(R.icop_d), A.c_dhw);
        A.c_ns = A.c_ns + e_ns;
        A.c_b = A.c_b + eb_b;
    }
    const F net = OUT ? zero : vfma(A.c_cool + A.c_heat + A.c_dhw + A.c_ns + A.c_b, splat<F>(B.r), splat<F>(R.sol));
    O.net = net; O.cost = net * R.price; O.emission = vmax(zero, net * R.carbon);
    if constexpr (DETAIL) {
        O.eb = eb_b;
        O.cool_dem = e_cool + vabs(vmin(eb_cs, zero));          // building.py:1435-1437
        O.heat_dem = e_heat + vabs(vmin(eb_hs, zero));
        O.dhw_dem = e_dhw + vabs(vmin(eb_ds, zero));
        // what Device.electricity_consumption reports: accumulator * time_step_ratio (energy_model.py:118)
        O.c_cool = A.c_cool * B.r; O.c_heat = A.c_heat * B.r; O.c_dhw = A.c_dhw * B.r; O.c_ns = A.c_ns * B.r;
        // evaluate()'s baseline: remove what the storages did (building.py:345-366, 413-463) and, for dynamics
        // buildings, add back the ideal-vs-delivered load difference (building.py:2877-2905)
        O.se_cool = eb_cs * R.icop_c; O.se_heat = eb_hs * R.icop_h; O.se_dhw = eb_ds * R.icop_d;      // *_storage_electricity_consumption (building.py:413-457)
        F base = net - vfma(A.c_b, splat<F>(B.r), vfma(eb_ds, splat<F>(R.icop_d), vfma(eb_hs, splat<F>(R.icop_h), eb_cs * R.icop_c)));
        O.net_ws = base;
        // (sic, building.py:2893-2898: the heating difference of every step is converted with ONE COP, the episode's last row's)
        if (B.flags & CLF_DYNAMICS) base = base + vfma(splat<F>(R.heat) - heat_dem, splat<F>(heat_hp ? R.icop_h_eval : B.t0_iheat_div), (splat<F>(R.cool) - cool_dem) * R.icop_c);
        O.base_net = base;
        O.expected = cool_dem + heat_dem + R.dhw + R.nsl;
        O.served = e_cool + vmax(-eb_cs, zero) + e_heat + vmax(-eb_hs, zero) + e_dhw + vmax(-eb_ds, zero) + e_ns;
    }
}

// Per-building reward from the unit's own quantities (reward_function.py:65-214); MARL is finished by the caller.
template <typename F>
CL_DEV F unit_reward(int kind, const FP& B, const St<F>& S, F net) {
    const F zero = splat<F>(0.0f);
    switch (kind) {
    case CLR_INDEPENDENT_SAC: return vmin(-net, zero);
    case CLR_SOLAR_PENALTY: {
        const F sg = vsel(net > 0.0f, splat<F>(1.0f), vsel(net < 0.0f, splat<F>(-1.0f), zero)), an = vabs(net);
        const uint32_t* __restrict__ f = B.f;
        F rw = cl::pw(f, (CLP_F_BATT - CLP_F_FIRST) + 2) > CL_ZDP ? -vfma(sg, S.soc, splat<F>(1.0f)) * an : zero;
        rw = rw + (cl::pw(f, (CLP_F_TANK - CLP_F_FIRST) + 0) > CL_ZDP ? -vfma(sg, S.cs, splat<F>(1.0f)) * an : zero);
        rw = rw + (cl::pw(f, (CLP_F_TANK - CLP_F_FIRST) + 8) > CL_ZDP ? -vfma(sg, S.hs, splat<F>(1.0f)) * an : zero);
        rw = rw + (cl::pw(f, (CLP_F_TANK - CLP_F_FIRST) + 16) > CL_ZDP ? -vfma(sg, S.ds, splat<F>(1.0f)) * an : zero);
        return rw;
    }
    case CLR_MARL: case CLR_EV: return net;   // placeholder, finished with the district sum
    default: {
        const F m = vmax(net, zero);
        return B.rw_exponent == 1.0f ? -m : -vpow(m, B.rw_exponent);
    }
    }
}

template <typename F>
CL_DEV F marl_partial(F net) {       // sign(-net) * 0.01 * net^2 (cl::marl_reward with district net 1)
    const F sg = vsel(net < 0.0f, splat<F>(1.0f), vsel(net > 0.0f, splat<F>(-1.0f), splat<F>(0.0f)));
    return sg * 0.01f * net * net;
}

}  // namespace clv

namespace {

// (OBS, round 6) cl_step_observe_f32 on thermal / outage districts: the compact observation of the NEXT row -- every column an affine map of a plane
// value the wave that stepped the building still holds (battery / tank states of charge, net, reward) -- written into the workgroup's
// [envs][pitch] LDS tile and streamed out behind the district reduction, like the battery + PV launch does (lean_step_body).  `row0`: the first of
// the lane's envs inside the tile; `ts_delta`: the env block's episode offset (cl_dims.env_row0).  The list is wave-uniform (kernel arguments).
template <typename F>
CL_DEV void obs_fill(const ObsFusedArgs& of, float* __restrict__ tile, int b, int row0, int ts_delta, const clv::St<F>& S, F net, F rw) {
    const float* __restrict__ trow = of.table + (size_t)(of.row + ts_delta) * of.n_cols;
    for (int d = of.start[b]; d < of.start[b + 1]; ++d) {
        const int col = of.deps[d].col, src = of.deps[d].src;
        const float scale = of.deps[d].scale, base = trow[col];
        const int pl = (src >> 20) & 0xFF;
        F v;
        if ((src >> 28) == CLOB_KIND_OUT) v = pl == CLO_NET ? net : rw;
        else v = pl == CLS_B_SOC ? S.soc : pl == CLS_B_EFF ? S.eff : pl == CLS_B_DEGCAP ? S.degcap : pl == CLS_CS_SOC ? S.cs : pl == CLS_HS_SOC ? S.hs : S.ds;
        if constexpr (clv::Tr<F>::N == 1) tile[(size_t)row0 * of.pitch + col] = fmaf(v, scale, base);
        else {
#pragma unroll
            for (int i = 0; i < clv::Tr<F>::N; ++i) tile[(size_t)(row0 + i) * of.pitch + col] = fmaf(v[i], scale, base);
        }
    }
}

// ... pad columns zeroed (like cl_observe_f32) by the whole workgroup before the barrier, the tile streamed out in 16-byte stores after it
CL_DEV void obs_pad(const ObsFusedArgs& of, float* __restrict__ tile, int rows) {
    const int np = of.pitch - of.n_cols;
    for (int i = threadIdx.x; i < rows * np; i += blockDim.x) tile[(size_t)(i / np) * of.pitch + of.n_cols + i % np] = 0.0f;
}
CL_DEV void obs_flush(const ObsFusedArgs& of, const float* __restrict__ tile, int env_first, int rows) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* src4 = reinterpret_cast<const f4*>(tile);
    f4* dst4 = reinterpret_cast<f4*>(of.obs + (size_t)env_first * of.pitch);
    const int total4 = rows * of.pitch / 4;                           // (pitch % 4 == 0: host)
    for (int q = threadIdx.x; q < total4; q += blockDim.x) __builtin_nontemporal_store(src4[q], dst4 + q);
}

// What a wave loads for one building: issued as one batch, ahead of the previous building's arithmetic.
template <typename F>
struct FullIn {
    F soc, eff, deg, cs, hs, ds, a_cs, a_hs, a_ds, a_es, a_cd, a_hd;
    uint32_t flags;
};

// Plane accesses are written element by element on `float` lvalues (the load / store vectoriser fuses them back into one
// dwordx2 access): a store through a float2 lvalue is typed "may alias anything", after which every parameter / row read of the
// loop stops being a scalar load and comes back as a uniform VECTOR load -- 90 VGPRs of wave-uniform data.
template <int VEC>
CL_DEV typename Vec<VEC>::type full_load(const float* __restrict__ p) {
    if constexpr (VEC == 1) return p[0];
    else {
        const float* __restrict__ q = static_cast<const float*>(__builtin_assume_aligned(p, 4 * VEC));
        typename Vec<VEC>::type v;
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = q[i];
        return v;
    }
}

template <int VEC>
CL_DEV typename Vec<VEC>::type full_action(const StepArgs& a, int col, int env0) {
    using F = typename Vec<VEC>::type;
    if (col < 0) return (F)(0.0f);
    const float* p = a.actions + (long long)col * a.act_stride_col;
    if (a.act_stride_env == 1) return full_load<VEC>(p + env0);
    F v;
    if constexpr (VEC == 1) v = p[(long long)env0 * a.act_stride_env];
    else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = p[(long long)(env0 + i) * a.act_stride_env];
    }
    return v;
}

// `f`: the building's CLP_F_* block; its first eight words (flags + the seven action columns) are read as one batch before
// any of them is tested, so the whole load phase costs one scalar round trip.
template <int VEC, bool LP = false>
CL_DEV void full_load_in(FullIn<typename Vec<VEC>::type>& in, const StepArgs& a, const uint32_t* __restrict__ f, int b, int env0, long long plane) {
    using F = typename Vec<VEC>::type;
    const uint32_t flags = clv::uword<LP>(f, 0);
    const int c_cs = (int)clv::uword<LP>(f, 1), c_hs = (int)clv::uword<LP>(f, 2), c_ds = (int)clv::uword<LP>(f, 3), c_es = (int)clv::uword<LP>(f, 4),
              c_cd = (int)clv::uword<LP>(f, 5), c_hd = (int)clv::uword<LP>(f, 6), a_coh = (int)clv::uword<LP>(f, 7);
    in.flags = flags;
    const long long off = (long long)b * a.n_env + env0;
    const F zero = (F)(0.0f), one = (F)(1.0f);
    in.soc = zero; in.eff = one; in.deg = zero; in.cs = zero; in.hs = zero; in.ds = zero;
    if (flags & CLF_BATTERY) {
        in.soc = full_load<VEC>(a.state + CLS_B_SOC * plane + off);
        in.eff = full_load<VEC>(a.state + CLS_B_EFF * plane + off);
        in.deg = full_load<VEC>(a.state + CLS_B_DEGCAP * plane + off);
    }
    if (flags & CLF_COOL_STO) in.cs = full_load<VEC>(a.state + CLS_CS_SOC * plane + off);
    if (flags & CLF_HEAT_STO) in.hs = full_load<VEC>(a.state + CLS_HS_SOC * plane + off);
    if (flags & CLF_DHW_STO) in.ds = full_load<VEC>(a.state + CLS_DS_SOC * plane + off);
    in.a_es = full_action<VEC>(a, c_es, env0);
    in.a_cs = full_action<VEC>(a, c_cs, env0);
    in.a_hs = full_action<VEC>(a, c_hs, env0);
    in.a_ds = full_action<VEC>(a, c_ds, env0);
    if (a_coh >= 0) {
        const F c = full_action<VEC>(a, a_coh, env0);
        in.a_cd = clv::vabs(clv::vmin(c, zero)); in.a_hd = clv::vabs(clv::vmax(c, zero));
    } else {
        in.a_cd = full_action<VEC>(a, c_cd, env0);
        in.a_hd = full_action<VEC>(a, c_hd, env0);
    }
}

template <int VEC, bool NT>
CL_DEV void full_store(float* __restrict__ p, typename Vec<VEC>::type v) {
    if constexpr (VEC == 1) {
        if constexpr (NT) __builtin_nontemporal_store(v, p);
        else p[0] = v;
    } else {
        float* __restrict__ q = static_cast<float*>(__builtin_assume_aligned(p, 4 * VEC));
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            if constexpr (NT) __builtin_nontemporal_store(v[i], q + i);
            else q[i] = v[i];
        }
    }
}

template <int VEC>
CL_DEV void full_accumulate(float (&q)[VEC], typename Vec<VEC>::type v) {
    if constexpr (VEC == 1) q[0] += v;
    else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) q[i] += v[i];
    }
}

// Thermal / outage districts, one env step.  Same 2-D tile as cl_step_kernel (a workgroup = 64 * VEC envs x a chunk of
// buildings, wave w advances buildings w, w + nw, ...; district sums through LDS in a fixed order), VEC = 1 or 2.
// MAXT = largest workgroup the instantiation is launched with: 1024 threads cap the kernel at 128 VGPRs (four waves per SIMD),
// which the two-env pack does not fit without spilling; districts of up to nine buildings per workgroup row run the 576-thread one.
// WPE = waves per SIMD the register allocation must leave room for: 9 buildings x 65 536 envs at two envs per lane are 18 waves
// per CU (4.5 per SIMD) -- at 101 VGPRs only one 9-wave workgroup fits a CU and the launch runs in two generations (11.5 us);
// capped at 96 VGPRs (five waves per SIMD) both workgroups are resident at once.
// (Measured and dropped, MI355X, 2020 schema 9 x 65 536, profiles/r02_thermal_sweep.log: issuing the state / action loads of a
//  wave's NEXT building before computing the current one -- 9.6 vs 8.8 us; one wave per SIMD walking all nine buildings, with or
//  without that prefetch -- 18 us, i.e. 2 us per building of which 0.45 us is arithmetic: the scalar parameter round trips are
//  what a lone wave cannot hide, so the launch wants several waves per SIMD rather than a deeper per-wave pipeline.  Round 2, with
//  the ISA checked this time -- the first version's loop counter had gone to a VGPR, which turns every parameter read into a vector
//  load; with the `live` test outside the loop it stays scalar: the same prefetch on the C4 shard, 1024 x 1024 in 32-building
//  chunks, where every wave of a SIMD is in the same phase: 16.4 vs 16.2 us without.  What a wave waits for between two buildings
//  is the scalar parameter chain of the next one, not its plane loads.)
// LP: the parameter blocks and time-series rows of the workgroup's buildings are staged in LDS by one cooperative round of vector
// loads (CL_LP_WORDS words per building behind the reduction area) instead of arriving through each wave's chain of dependent
// scalar loads -- flags -> head + row -> one tank after the other -> battery -- which is what a wave of the building-chunked launch
// spends its time on (scripts/wave_timeline.py, 1024 x 1024: 2.1 us from entry to the first building's inputs, 1.9 us between
// the first building's stores and the second one's inputs, next to 2 x 2.0 us of arithmetic).  As VGPR operands the parameters
// also stop costing a v_mov per two-scalar instruction, and the SGPR file no longer spills.
constexpr int CL_LP_WORDS = (CLP_F_LAST - CLP_F_FIRST + 1) + CL_NF;      // 64 + 16

// KPI (round 3, cl_step_full_kpi_kernel): the wave that steps a building also updates the building's streaming KPI accumulators
// (CLD_KPI) -- net, baseline, expected and served energy are in its registers, the ten (outage steps: twelve) accumulator loads are
// issued next to the state loads, long before the arithmetic needs anything, and their stores leave with the plane stores -- and the
// workgroup feeds the two district series after the district reduction.  Replaces the step launch with the CLD_DETAIL_MIN planes + the
// cl_kpi_kernel launch that read them back (9 x 65 536: 29.1 us for the pair).  Same values in the same order as cl_kpi_kernel: bit-identical.
// Measured at 9 x 65 536 (scripts/kpi_cost_probe.py, profiles/r03_kpi_in_step_probe.log): 18.8 us with four waves per workgroup -- 4096
// waves, what the chip holds at once at this kernel's 119 registers; the step-only default of five waves (two generations) 23.6 us, one
// building per wave 23.8 us.  Tried and slower: the accumulator loads at their point of use with 88 registers and five waves per SIMD
// (19.5 us at best); wave specialisation -- two extra waves per workgroup that fetch the accumulators at entry, meet the step waves at the
// reduction's barrier, where those have parked net / baseline / expected / served in LDS, and add and store while the step waves reduce:
// 22.7 - 34 us (every accumulator store of a workgroup leaves after its slowest step wave, nothing overlaps them any more); plain instead of
// non-temporal accumulator stores (they are read back by the next step): no difference at 65 536 or 262 144 envs; two envs per lane
// (576-thread workgroups, three waves per SIMD, 168 registers with a few spills; the vector ALUs are ~40 % busy at one env per lane
// and the pack halves their work): bit-identical and 24 - 29 us against 17.7 us at 9 x 65 536, slower at every size tried -- the
// waves in flight, not the instruction count, carry this kernel.
template <int VEC, bool DETAIL, bool LP, bool NT, bool KPI, int PREC = 0, bool OBS = false>
CL_DEV void full_step_body(const StepArgs& a, [[maybe_unused]] const ObsFusedArgs* of = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [nw][NQ][64*VEC], then (LP) [buildings of the workgroup][CL_LP_WORDS] or (KPI) [n_bldg][64*VEC] baselines or (OBS) [64*VEC][pitch]
    static_assert(!KPI || (DETAIL && !LP && VEC == 1), "the KPI epilogue is written for one env per lane, with the baseline / expected / served values of the detail unit");
    static_assert(!OBS || (!LP && !KPI && !DETAIL), "the fused observation tile shares the LDS region behind the reduction rows");
    using F = typename Vec<VEC>::type;
    constexpr int TILE = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool SWAP = LP && CL_SWAP_GRID;             // (the LP instantiations are launched with grid = (building chunks, env tiles): district_reduce's note)
    const int bx = SWAP ? blockIdx.y : blockIdx.x, by = SWAP ? blockIdx.x : blockIdx.y;
    const int env0 = bx * TILE + lane * VEC;
    const bool live = env0 < a.n_env;                     // n_env % 4 == 0 is enforced on the host
    const long long plane = (long long)a.n_bldg * a.n_env;
    const int rkind = (a.flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT;
    const bool first = (a.flags & CLD_REF_T0_QUIRK) && a.t == 0;

    float q_net[VEC], q_cost[VEC], q_em[VEC], q_rw[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) q_net[i] = q_cost[i] = q_em[i] = q_rw[i] = 0.0f;

    const int b_lo = by * a.b_chunk;
    const int b_hi = min(a.n_bldg, b_lo + a.b_chunk);
    const bool marl_partial = rkind == CLR_MARL && a.n_chunks > 1;
    const int ts_row = a.t + (a.env_row0 ? a.env_row0[(bx * TILE) / CL_ROW0_BLOCK] : 0);
    CL_TRACE_DECL;
    CL_TRACE_ENTRY(0);
    CL_TRACE_CYCLES_ENTRY(4);     // shader-clock cycles at entry (slot 12: at the end) -- gives the clock the launch ran at
    [[maybe_unused]] int tr_i = 0;
    [[maybe_unused]] uint32_t* stage = reinterpret_cast<uint32_t*>(lds + (size_t)a.nw * NQ * TILE);
    constexpr bool FOLDK = LP && VEC == 2;               // the C4 shard's kernel: district_reduce<.., FOLD>
    // ... whose 128 registers (16 waves per workgroup) do not hold two envs' unit, the next building's inputs, the fold's value AND the eight
    // district accumulators: the accumulators live in the wave's own LDS row instead (zeroed here, read - add - written after every building:
    // the same additions in the same order as in registers -- (0 + O1) + O2 -- and nobody else touches the row before district_reduce's barrier)
    constexpr bool QLDS = LP;
    [[maybe_unused]] float* qrow = lds + (size_t)w * NQ * TILE + lane * VEC;
    if constexpr (QLDS) {
        const F zero = (F)(0.0f);
#pragma unroll
        for (int q = 0; q < NQ; ++q) full_store<VEC, false>(qrow + q * TILE, zero);
    }
    [[maybe_unused]] float fold_prev = 0.0f;
    [[maybe_unused]] bool fold_issued = false, folded = false;
    // (deferred finish) [64 chunks][16 district sums], behind the staged parameter blocks
    [[maybe_unused]] float* lds_fold = lds + (size_t)a.nw * NQ * TILE + (LP ? (size_t)a.b_chunk * CL_LP_WORDS : 0);
    // (LP) The wave's FIRST building does not wait for the staging round trip: its eight header words (flags + action columns) come through
    // scalar loads from the parameter table itself and its plane / action loads are in flight before the staging loads are (round 4; as the
    // loop's first statement they sat behind staging loads -> LDS writes -> barrier: one more dependent round trip at the head of every
    // workgroup of a single-generation launch).
    [[maybe_unused]] FullIn<F> early;
    [[maybe_unused]] bool have_early = false;
    if constexpr (LP) {
        if (live && b_lo + w < b_hi) {
            full_load_in<VEC, false>(early, a, a.params + (long long)(b_lo + w) * CL_NP + CLP_F_FIRST, b_lo + w, env0, plane);
            have_early = true;
            if constexpr (FOLDK) { fold_prev = fold_prefetch<TILE>(a, w, lane, plane, bx, by); fold_issued = true; }
        }
    }
    if constexpr (LP) {
        static_assert(!DETAIL, "the evaluate()-time COP row of the detail planes is not staged");
        const int n_words = (b_hi - b_lo) * CL_LP_WORDS;
        const uint32_t* __restrict__ tsw = reinterpret_cast<const uint32_t*>(a.ts);
        for (int i = threadIdx.x; i < n_words; i += blockDim.x) {
            const int j = i / CL_LP_WORDS, k = i - j * CL_LP_WORDS, b = b_lo + j;
            stage[i] = k < CL_LP_WORDS - CL_NF ? a.params[(long long)b * CL_NP + CLP_F_FIRST + k]
                                               : tsw[((long long)ts_row * a.n_bldg + b) * CL_NF + (k - (CL_LP_WORDS - CL_NF))];
        }
        __syncthreads();
    }
    [[maybe_unused]] float* hand = lds + (size_t)a.nw * NQ * TILE;        // (KPI) [n_bldg][TILE]: the buildings' baselines of this step
    // one building of the wave: `cur` = its inputs (in flight); everything else of the unit
    auto building = [&](const int b, FullIn<F>& cur) {
        {
            const uint32_t* __restrict__ f = LP ? stage + (b - b_lo) * CL_LP_WORDS : a.params + (long long)b * CL_NP + CLP_F_FIRST;
            if constexpr (FOLDK) {
                // (deferred finish: this wave's share of the previous step's chunk sums, issued behind the first building's plane loads --
                //  fold_prefetch's note)
                if (!fold_issued) { fold_prev = fold_prefetch<TILE>(a, w, lane, plane, bx, by); fold_issued = true; }
            }
            clv::FP B;
            clv::load_fp<LP>(B, f);
            [[maybe_unused]] const uint32_t* __restrict__ grow = PREC == 2 ? a.params + (long long)b * CL_NP : nullptr;
            cl::Row R;
            cl::load_row_scalar<true>(R, LP ? reinterpret_cast<const float*>(f + (CL_LP_WORDS - CL_NF)) : a.ts + ((long long)ts_row * a.n_bldg + b) * CL_NF, B.flags,
                                      DETAIL ? a.ts + ((long long)(ts_row - a.t + a.n_steps - 1) * a.n_bldg + b) * CL_NF : nullptr);
            const long long off = (long long)b * a.n_env + env0;
            // (KPI) every accumulator is loaded before anything is stored, here, with the inputs in flight (cl_kpi_kernel's note: a `+=`
            // per accumulator is a chain of dependent round trips)
            [[maybe_unused]] float kv[CL_NKB];
            if constexpr (KPI) {
                const float* __restrict__ k = a.kpi_bldg + off;
#pragma unroll
                for (int p = 0; p < CL_NKB; ++p) kv[p] = (R.outage || (p != CLK_UNSERVED_OUTAGE && p != CLK_EXPECTED_OUTAGE)) ? k[p * plane] : 0.0f;
            }
            CL_TRACE_INPUTS(1 + 4 * tr_i, cur);
            clv::St<F> S = {cur.soc, cur.eff, cur.deg, cur.cs, cur.hs, cur.ds};
            const clv::Ac<F> act = {cur.a_cs, cur.a_hs, cur.a_ds, cur.a_es, cur.a_cd, cur.a_hd};
            clv::Ou<F> O;
            if (R.outage) clv::unit_step<F, true, DETAIL, PREC>(B, R, a.t, first, act, S, O, grow);
            else clv::unit_step<F, false, DETAIL, PREC>(B, R, a.t, first, act, S, O, grow);
            F rw = clv::unit_reward<F>(rkind, B, S, O.net);
            if constexpr (OBS) obs_fill<F>(*of, lds + (size_t)a.nw * NQ * TILE, b, lane * VEC, ts_row - a.t, S, O.net, rw);
            CL_TRACE_AFTER(2 + 4 * tr_i, rw);
            CL_TRACE_AFTER(2 + 4 * tr_i, S.soc);
            if constexpr (FOLDK) {
                if (!folded && b + a.nw >= b_hi) { fold_stash(a, lds_fold, w, lane, fold_prev); folded = true; }      // ... and parked in LDS before the LAST building's stores
            }
            if (B.flags & CLF_BATTERY) {
                full_store<VEC, NT>(a.state + CLS_B_SOC * plane + off, S.soc);
                full_store<VEC, NT>(a.state + CLS_B_EFF * plane + off, S.eff);
                full_store<VEC, NT>(a.state + CLS_B_DEGCAP * plane + off, S.degcap);
            }
            if (B.flags & CLF_COOL_STO) full_store<VEC, NT>(a.state + CLS_CS_SOC * plane + off, S.cs);
            if (B.flags & CLF_HEAT_STO) full_store<VEC, NT>(a.state + CLS_HS_SOC * plane + off, S.hs);
            if (B.flags & CLF_DHW_STO) full_store<VEC, NT>(a.state + CLS_DS_SOC * plane + off, S.ds);
            full_store<VEC, NT>(a.out_bldg + CLO_NET * plane + off, O.net);
            if (rkind != CLR_MARL) full_store<VEC, NT>(a.out_bldg + CLO_REWARD * plane + off, rw);
            if constexpr (KPI) {
                const float net = O.net, base = O.base_net, ex = O.expected, sv = O.served;
                kv[CLK_C_POS] += fmaxf(net, 0.0f);
                kv[CLK_C_NET] += net;
                kv[CLK_C_EMISSION] += fmaxf(net * R.carbon, 0.0f);
                kv[CLK_C_COST] += fmaxf(net * R.price, 0.0f);
                kv[CLK_B_POS] += fmaxf(base, 0.0f);
                kv[CLK_B_NET] += base;
                kv[CLK_B_EMISSION] += fmaxf(base * R.carbon, 0.0f);
                kv[CLK_B_COST] += fmaxf(base * R.price, 0.0f);
                kv[CLK_UNSERVED_OUTAGE] += ex - sv; kv[CLK_EXPECTED_OUTAGE] += ex;
                kv[CLK_UNSERVED_ALL] += ex - sv;
                kv[CLK_EXPECTED_ALL] += ex;
                float* __restrict__ k = a.kpi_bldg + off;
#pragma unroll
                for (int p = 0; p < CL_NKB; ++p)
                    if (R.outage || (p != CLK_UNSERVED_OUTAGE && p != CLK_EXPECTED_OUTAGE)) full_store<1, NT>(k + p * plane, kv[p]);
                hand[(size_t)b * TILE + lane] = base;                                      // the district baseline series is summed below
            }
            if (DETAIL && (!KPI || (a.flags & CLD_WRITE_DETAIL))) {
                // what another kernel of the path reads: the KPI pass (baseline, expected, served) and the LSTM stage (delivered demands)
                full_store<VEC, NT>(a.out_bldg + CLO_COOL_DEM * plane + off, O.cool_dem);
                full_store<VEC, NT>(a.out_bldg + CLO_HEAT_DEM * plane + off, O.heat_dem);
                full_store<VEC, NT>(a.out_bldg + CLO_BASE_NET * plane + off, O.base_net);
                full_store<VEC, NT>(a.out_bldg + CLO_EXPECTED * plane + off, O.expected);
                full_store<VEC, NT>(a.out_bldg + CLO_SERVED * plane + off, O.served);
                if (!(a.flags & CLD_DETAIL_MIN)) {                           // the series of evaluate() / the observations / the parity tests
                    full_store<VEC, NT>(a.out_bldg + CLO_B_EB * plane + off, O.eb);
                    full_store<VEC, NT>(a.out_bldg + CLO_DHW_DEM * plane + off, O.dhw_dem);
                    full_store<VEC, NT>(a.out_bldg + CLO_C_COOL * plane + off, O.c_cool);
                    full_store<VEC, NT>(a.out_bldg + CLO_C_HEAT * plane + off, O.c_heat);
                    full_store<VEC, NT>(a.out_bldg + CLO_C_DHW * plane + off, O.c_dhw);
                    full_store<VEC, NT>(a.out_bldg + CLO_C_NSL * plane + off, O.c_ns);
                    full_store<VEC, NT>(a.out_bldg + CLO_NET_WS * plane + off, O.net_ws);
                    full_store<VEC, NT>(a.out_bldg + CLO_SE_COOL * plane + off, O.se_cool);
                    full_store<VEC, NT>(a.out_bldg + CLO_SE_HEAT * plane + off, O.se_heat);
                    full_store<VEC, NT>(a.out_bldg + CLO_SE_DHW * plane + off, O.se_dhw);
                }
            }
            // multi-chunk MARL: accumulate sign(-net) * 0.01 * net^2; cl_finish_kernel scales by max(0, district net)
            if constexpr (QLDS) {
                static_assert(CLQ_NET == 0 && CLQ_COST == 1 && CLQ_EMISSION == 2 && CLQ_REWARD == 3, "row order of the LDS accumulators");
                const F add[NQ] = {O.net, O.cost, O.emission, marl_partial ? clv::marl_partial<F>(O.net) : rw};
#pragma unroll
                for (int q = 0; q < NQ; ++q) full_store<VEC, false>(qrow + q * TILE, full_load<VEC>(qrow + q * TILE) + add[q]);
            } else {
                full_accumulate<VEC>(q_net, O.net); full_accumulate<VEC>(q_cost, O.cost); full_accumulate<VEC>(q_em, O.emission);
                full_accumulate<VEC>(q_rw, marl_partial ? clv::marl_partial<F>(O.net) : rw);
            }
            CL_TRACE_AFTER(3 + 4 * tr_i, q_rw[0]);          // stores issued
#ifdef CL_TRACE
            if (tr_i < 2) ++tr_i;
#endif
        }
    };
    if (live) {
        int b = b_lo + w;
        if (LP && have_early) { building(b, early); b += a.nw; }              // (peeled: the loop body proper never sees `early`)
        for (; b < b_hi; b += a.nw) {
            FullIn<F> cur;
            full_load_in<VEC, LP>(cur, a, LP ? stage + (b - b_lo) * CL_LP_WORDS : a.params + (long long)b * CL_NP + CLP_F_FIRST, b, env0, plane);
            building(b, cur);
        }
    }
    CL_TRACE_AFTER(13, q_net[0]);
    [[maybe_unused]] KpiSeries base_pre;
    if constexpr (KPI) {
        // (the baseline series' accumulators of this lane's env: fetched before the reduction's barriers, by the LAST wave -- wave 0 carries the control series)
        if (w == a.nw - 1 && live) kpi_series_fetch(base_pre, a.kpi_env + (long long)CLKE_PER_COND * a.n_env + env0, a.n_env);
    }
    if constexpr (FOLDK) {
        if (!folded) {
            if (!fold_issued) fold_prev = fold_prefetch<TILE>(a, w, lane, plane, bx, by);
            fold_stash(a, lds_fold, w, lane, fold_prev);
        }
    }
    if constexpr (OBS) obs_pad(*of, lds + (size_t)a.nw * NQ * TILE, TILE);
    district_reduce<VEC, false, FOLDK, KPI, QLDS, SWAP>(a, lds, w, lane, env0, live, plane, rkind, q_net, q_cost, q_em, q_rw, a.nw, lds_fold);      // (LP, two envs per lane: the C4 shard's kernel, which may fold its chunk sums)
    if constexpr (OBS) obs_flush(*of, lds + (size_t)a.nw * NQ * TILE, bx * TILE, min(TILE, a.n_env - bx * TILE));      // (district_reduce's first barrier came after every wave's tile writes; never chunked: host)
    if constexpr (KPI) {
        // baseline district series: the per-building baselines in cl_kpi_kernel's association (16 strided partial sums, added in order).
        // (district_reduce's barriers came after every wave's LDS writes; MARL's extra sweep leaves this region alone.)
        if (w == a.nw - 1 && live) {
            const float* __restrict__ bl = hand + lane;
            float base = 0.0f;
            for (int k = 0; k < 16; ++k) {
                float s = 0.0f;
                for (int b = k; b < a.n_bldg; b += 16) s += bl[(size_t)b * TILE];
                base += s;
            }
            kpi_series_apply(a.kpi_env + (long long)CLKE_PER_COND * a.n_env + env0, a.n_env, a.t, base, base_pre);
        }
    }
    CL_TRACE_FLUSH();
}

template <int VEC, bool DETAIL, int MAXT, int WPE, bool LP, bool NT>
__global__ void __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(WPE))) cl_step_full_kernel(const StepArgs a) {
    full_step_body<VEC, DETAIL, LP, NT, false>(a);
}

// CLD_F64_CHAIN: the same kernel around cl::battery_charge_chain (the float64 soc chain; csrc/cl_unit.h)
template <int VEC, bool DETAIL, int MAXT, int WPE, bool LP, bool NT>
__global__ void __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(WPE))) cl_step_full_chain_kernel(const StepArgs a) {
    full_step_body<VEC, DETAIL, LP, NT, false, 2>(a);
}

// ... with the compact observation of the next row written by the same launch (cl_step_observe_f32; one env per lane, one workgroup row)
template <int PREC, bool NT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4))) cl_step_full_obs_kernel(const StepArgs a, const ObsFusedArgs of) {
    full_step_body<1, false, false, NT, false, PREC, true>(a, &of);
}

template <bool NT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4))) cl_step_full_kpi_kernel(const StepArgs a) {
    full_step_body<1, true, false, NT, true>(a);
}

// Thermal / outage districts, several env tiles per workgroup ("items" = (env tile, building) pairs dealt to the waves in order:
// wave w takes items w, w + nw, ...).  With nine buildings a workgroup of one 128-env tile has nine units of work for four SIMDs;
// two tiles give 18 items to 16 waves -- SIMD loads 5, 5, 4, 4, one workgroup per CU, all resident at once, and only two waves walk
// two items.  Every item parks its four district partials in its own LDS row; the sums are then formed per (tile, quantity, env)
// in building order (the reference's order, citylearn.py:1909-1918).
// (Round 3, tried: a wave that walks two items fetching the second item's planes before it stores the first one's -- the stores otherwise
//  fence the loads behind them.  102 instead of 77 registers and SLOWER: 9 x 65 536 8.44 vs 7.86 us, 9 x 262 144 29.9 vs 29.2 us.)
template <int VEC, bool NT, int PREC, bool OBS = false>
CL_DEV void full_tp_body(const StepArgs& a, const int tp, [[maybe_unused]] const ObsFusedArgs* of = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [tp * n_bldg][NQ][64*VEC], then [tp][64*VEC], then (OBS) [tp * 64*VEC][pitch]
    using F = typename Vec<VEC>::type;
    constexpr int TILE = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long plane = (long long)a.n_bldg * a.n_env;
    const int rkind = (a.flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT;
    const bool first = (a.flags & CLD_REF_T0_QUIRK) && a.t == 0;
    const int tile0 = blockIdx.x * tp * TILE;
    const int ts_row = a.t + (a.env_row0 ? a.env_row0[tile0 / CL_ROW0_BLOCK] : 0);       // tp * TILE <= CL_ROW0_BLOCK (host)
    const int n_items = tp * a.n_bldg;
    const bool marl = rkind == CLR_MARL;
    [[maybe_unused]] float* otile = lds + ((size_t)n_items * NQ + tp) * TILE;
    for (int j = w; j < n_items; j += a.nw) {
        const int h = j / a.n_bldg, b = j - h * a.n_bldg;
        const int env0 = tile0 + h * TILE + lane * VEC;
        const bool live = env0 < a.n_env;
        F net = (F)(0.0f), cost = (F)(0.0f), em = (F)(0.0f), rws = (F)(0.0f);
        if (live) {
            const uint32_t* __restrict__ f = a.params + (long long)b * CL_NP + CLP_F_FIRST;
            FullIn<F> cur;
            full_load_in<VEC>(cur, a, f, b, env0, plane);
            clv::FP B;
            clv::load_fp(B, f);
            [[maybe_unused]] const uint32_t* __restrict__ grow = PREC == 2 ? a.params + (long long)b * CL_NP : nullptr;
            cl::Row R;
            cl::load_row_scalar<true>(R, a.ts + ((long long)ts_row * a.n_bldg + b) * CL_NF, B.flags, nullptr);
            clv::St<F> S = {cur.soc, cur.eff, cur.deg, cur.cs, cur.hs, cur.ds};
            const clv::Ac<F> act = {cur.a_cs, cur.a_hs, cur.a_ds, cur.a_es, cur.a_cd, cur.a_hd};
            clv::Ou<F> O;
            if (R.outage) clv::unit_step<F, true, false, PREC>(B, R, a.t, first, act, S, O, grow);
            else clv::unit_step<F, false, false, PREC>(B, R, a.t, first, act, S, O, grow);
            const F rw = clv::unit_reward<F>(rkind, B, S, O.net);
            if constexpr (OBS) obs_fill<F>(*of, otile, b, h * TILE + lane * VEC, ts_row - a.t, S, O.net, rw);
            const long long off = (long long)b * a.n_env + env0;
            if (B.flags & CLF_BATTERY) {
                full_store<VEC, NT>(a.state + CLS_B_SOC * plane + off, S.soc);
                full_store<VEC, NT>(a.state + CLS_B_EFF * plane + off, S.eff);
                full_store<VEC, NT>(a.state + CLS_B_DEGCAP * plane + off, S.degcap);
            }
            if (B.flags & CLF_COOL_STO) full_store<VEC, NT>(a.state + CLS_CS_SOC * plane + off, S.cs);
            if (B.flags & CLF_HEAT_STO) full_store<VEC, NT>(a.state + CLS_HS_SOC * plane + off, S.hs);
            if (B.flags & CLF_DHW_STO) full_store<VEC, NT>(a.state + CLS_DS_SOC * plane + off, S.ds);
            full_store<VEC, NT>(a.out_bldg + CLO_NET * plane + off, O.net);
            if (!marl) full_store<VEC, NT>(a.out_bldg + CLO_REWARD * plane + off, rw);
            net = O.net; cost = O.cost; em = O.emission; rws = rw;
        }
        float* row = lds + (size_t)j * NQ * TILE + lane * VEC;
        full_store<VEC, false>(row + CLQ_NET * TILE, net); full_store<VEC, false>(row + CLQ_COST * TILE, cost);
        full_store<VEC, false>(row + CLQ_EMISSION * TILE, em); full_store<VEC, false>(row + CLQ_REWARD * TILE, rws);
    }
    if constexpr (OBS) obs_pad(*of, otile, tp * TILE);
    __syncthreads();
    if constexpr (OBS) obs_flush(*of, otile, tile0, min(tp * TILE, a.n_env - tile0));      // (the workgroup's tiles are consecutive envs: one contiguous block of rows)
    float* dnet = lds + (size_t)n_items * NQ * TILE;                  // [tp][TILE]: district net (MARL)
    auto sum_rows = [&](int q_lo, int q_hi) {
        const int nq = q_hi - q_lo;
        for (int o = threadIdx.x; o < tp * nq * TILE; o += blockDim.x) {
            const int h = o / (nq * TILE), r = o - h * nq * TILE, q = q_lo + r / TILE, e = r % TILE;
            float s = 0.0f;
            for (int b = 0; b < a.n_bldg; ++b) s += lds[((size_t)(h * a.n_bldg + b) * NQ + q) * TILE + e];
            const int env = tile0 + h * TILE + e;
            if (env < a.n_env) a.out_env[(long long)q * a.n_env + env] = s;
            if (marl && q == CLQ_NET) dnet[h * TILE + e] = s;
        }
    };
    static_assert(CLQ_NET == 0 && CLQ_REWARD == NQ - 1, "the MARL pass sums the reward rows separately");
    sum_rows(0, marl ? NQ - 1 : NQ);
    if (marl) {
        // MARL couples every building to the district net (reward_function.py:132-143): each wave revisits its items with the
        // finished net of their env tile, then the reward rows are summed
        __syncthreads();
        for (int j = w; j < n_items; j += a.nw) {
            const int h = j / a.n_bldg, b = j - h * a.n_bldg;
            const int env0 = tile0 + h * TILE + lane * VEC;
            float* row = lds + (size_t)j * NQ * TILE + lane * VEC;
            F rw;
            if constexpr (VEC == 1) rw = cl::marl_reward(row[CLQ_NET * TILE], dnet[h * TILE + lane]);
            else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) rw[i] = cl::marl_reward(row[CLQ_NET * TILE + i], dnet[h * TILE + lane * VEC + i]);
            }
            if (env0 < a.n_env) full_store<VEC, NT>(a.out_bldg + CLO_REWARD * plane + (long long)b * a.n_env + env0, rw);
            else rw = (F)(0.0f);
            full_store<VEC, false>(row + CLQ_REWARD * TILE, rw);
        }
        __syncthreads();
        sum_rows(NQ - 1, NQ);
    }
}

template <int VEC, int WPE, bool NT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(WPE))) cl_step_full_tp_kernel(const StepArgs a, const int tp) {
    full_tp_body<VEC, NT, 0>(a, tp);
}

// CLD_F64_CHAIN
template <int VEC, int WPE, bool NT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(WPE))) cl_step_full_tp_chain_kernel(const StepArgs a, const int tp) {
    full_tp_body<VEC, NT, 2>(a, tp);
}

// ... with the compact observation of the next row written by the same launch (cl_step_observe_f32)
template <int VEC, int PREC, bool NT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4))) cl_step_full_tp_obs_kernel(const StepArgs a, const int tp, const ObsFusedArgs of) {
    full_tp_body<VEC, NT, PREC, true>(a, tp, &of);
}

}  // namespace

#pragma clang fp contract(fast)
