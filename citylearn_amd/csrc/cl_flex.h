// cl_flex.h -- EV chargers, electric vehicles and washing machines; included by cl_kernels.hip.
//
// One launch per env step, BEFORE the building step kernel: it leaves, for every building that owns a charger or a
// washing machine, the electricity they drew this step (cl_flex_out planes) and the net-independent part of
// Electric_Vehicles_Reward_Function; the step kernel adds the load to the net and finishes the reward.
//
// Work decomposition: a wave owns 64 consecutive envs of ONE unit, so every table read is wave-uniform (scalar) and
// every plane access is one coalesced request.  Units are
//   [0, n_flex_bldg)         a building: its chargers in order, then its washing machines (the summation order of
//                            Building.update_variables, building.py:2657-2680), each charger advancing the EV it holds;
//   [n_flex_bldg, +n_ev)     an EV that no charger holds on this row: the arrival / drift rules of
//                            CityLearnEnv.simulate_unconnected_ev_soc (citylearn.py:1416-1474).
// An EV is advanced by exactly one unit per step (CLEV_CONNECTED decides which), so no two waves touch the same state.
//
// SoC bookkeeping.  The reference keeps soc[t] series that start at zero and are written only by Battery.charge or
// force_set_soc -- nothing carries a value forward.  ev_state's SoC plane holds the entry of the LAST step (soc[t-1]);
// the entry of step t starts from the host-packed rule of (row, EV) (0, an arrival SoC, or the drift of soc[t-1]) and
// is replaced by Battery.charge only when the connected charger's action is non-zero
// (electric_vehicle_charger.py:306, 331-334): a zero action on a connected EV leaves the rule value, usually 0.
#pragma once

namespace {

struct FlexArgs {
    cl_flex f;
    const float* __restrict__ actions;
    long long act_stride_col, act_stride_env;
    const int32_t* __restrict__ env_row0;
    int n_env, n_steps, t;
    unsigned env_offset;   // cl_dims.env_offset: keys the drift stream by the env's index in the whole batch
    bool want_reward;      // reward kind is CLR_EV: leave the K planes
    bool want_chargers;    // the step writes evaluate()'s baseline (CLD_WRITE_DETAIL): leave the chargers-only plane
};

CL_DEV float flex_action(const FlexArgs& a, int col, int env) {
    if (col < 0) return 0.0f;
    return a.actions[(long long)col * a.act_stride_col + (long long)env * a.act_stride_env];
}

// N(1, 0.2) multiplier of the SoC drift for (env, EV, t): Box-Muller on two Philox words; its own counter space
// (column bit 31 set) so it never collides with the rollout policy stream of cl_rollout.h.
CL_DEV float flex_drift_multiplier(unsigned long long seed, int env, int ev, int t) {
    const cl::U4 b = cl::philox_block(seed, (uint32_t)env, 0x80000000u | (uint32_t)ev, (uint32_t)t);
    const float u1 = 1.0f - cl::u01(b.w[0]), u2 = cl::u01(b.w[1]);      // u1 in (0, 1]
    return fmaf(0.2f, sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2), 1.0f);
}

// soc[t] before any charger action: the rule of (row, EV) applied to soc[t - 1] (citylearn.py:1416-1474, 1353-1414)
CL_DEV float flex_begin_soc(const FlexArgs& a, const float* __restrict__ ev_row, int row, int ev, int env, float prev) {
    const bool last = a.t + 1 >= a.n_steps;
    const float rule = ev_row[last ? CLEV_RULE_LAST : CLEV_RULE_STEP];
    if (rule >= 0.0f) return rule;
    if (rule == CLEV_ZERO) return 0.0f;
    const float m = a.f.drift ? a.f.drift[(long long)row * a.f.n_ev + ev] : flex_drift_multiplier(a.f.seed, (int)((unsigned)env + a.env_offset), ev, a.t);
    return fminf(fmaxf(prev * fminf(fmaxf(m, 0.6f), 1.4f), 0.0f), 1.0f);
}

// np.interp(x, xs, ys) over the n <= CL_CURVE_MAX points of a charger efficiency curve (end values outside the range)
CL_DEV float flex_curve(const uint32_t* __restrict__ cp, int n_slot, int x_slot, int y_slot, float x) {
    const int n = (int)cp[n_slot];
    float y = cl::pw(cp, y_slot);
    for (int k = 1; k < n; ++k) {                 // wave-uniform trip count
        const float x0 = cl::pw(cp, x_slot + k - 1), x1 = cl::pw(cp, x_slot + k);
        const float y0 = cl::pw(cp, y_slot + k - 1), y1 = cl::pw(cp, y_slot + k);
        const float seg = fmaf((x - x0) * cl::rcp(x1 - x0), y1 - y0, y0);
        y = x >= x1 ? y1 : (x > x0 ? seg : y);
    }
    return y;
}

// VEC consecutive envs per lane (float4 plane accesses at VEC = 4: a quarter of the waves walk the scalar table chain).
template <int VEC, bool NT>
__global__ void __launch_bounds__(256) cl_flex_kernel(const FlexArgs a) {
    constexpr int TILE = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int u = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    const int env0 = blockIdx.x * TILE + lane * VEC;
    const cl_flex& f = a.f;
    if (u >= f.n_flex_bldg + f.n_ev) return;
    const bool live = env0 < a.n_env;                     // n_env % 4 == 0 (check_dims): a lane's VEC envs are all in or all out
    const int row = a.t + (a.env_row0 ? a.env_row0[(blockIdx.x * TILE) / CL_ROW0_BLOCK] : 0);
    const long long ev_plane = (long long)f.n_ev * a.n_env;
    const bool coalesced = a.act_stride_env == 1;

    if (u >= f.n_flex_bldg) {
        // ---- an EV nobody holds on this row ----
        const int k = u - f.n_flex_bldg;
        const float* __restrict__ er = f.ev_ts + ((long long)row * f.n_ev + k) * CL_NEVF;
        if (a.t == 0 || er[CLEV_CONNECTED] != 0.0f || !live) return;     // t = 0: cl_flex_reset_f32 wrote soc[0]
        float* sp = f.ev_state + (long long)k * a.n_env + env0;
        float soc[VEC];
        vload<VEC>(soc, sp);
#pragma unroll
        for (int i = 0; i < VEC; ++i) soc[i] = flex_begin_soc(a, er, row, k, env0 + i, soc[i]);
        pstore<VEC, NT>(sp, soc);
        return;
    }

    // ---- a building: chargers, then washing machines ----
    // Phase 1: the slot headers (scalar; every address depends on (u, row) only).  Phase 2: every plane / action read of
    // every occupied slot, issued back to back BEFORE any store -- the compiler may not move a load above an earlier
    // store to a possibly aliasing plane, so interleaving load / compute / store per slot serialised one HBM round trip
    // per slot.  Phase 3: arithmetic and stores.
    const uint32_t* __restrict__ cp0 = f.charger_params + (long long)u * CL_MAXC * CL_NCP;
    const float* __restrict__ cr0 = f.charger_ts + ((long long)row * f.n_flex_bldg + u) * CL_MAXC * CL_NCF;
    const uint32_t* __restrict__ wp0 = f.wm_params + (long long)u * CL_MAXW * CL_NWP;
    const float* __restrict__ wr0 = f.wm_ts + ((long long)row * f.n_flex_bldg + u) * CL_MAXW * CL_NWF;
    float hdr[CL_MAXC], whdr[CL_MAXW];
    int ccol[CL_MAXC], wcol[CL_MAXW];
#pragma unroll
    for (int j = 0; j < CL_MAXC; ++j) { hdr[j] = cr0[j * CL_NCF + CLCT_EV]; ccol[j] = (int)cp0[j * CL_NCP + CLC_ACT_COL]; }
#pragma unroll
    for (int j = 0; j < CL_MAXW; ++j) { whdr[j] = wr0[j * CL_NWF + CLWT_OPEN]; wcol[j] = (int)wp0[j * CL_NWP]; }
    if (!live) return;
    float act[CL_MAXC][VEC], soc[CL_MAXC][VEC], ef[CL_MAXC][VEC], deg[CL_MAXC][VEC], winit[CL_MAXW][VEC], wact[CL_MAXW][VEC];
#pragma unroll
    for (int j = 0; j < CL_MAXC; ++j) {
        if (hdr[j] == CLCT_EMPTY) continue;
        if (ccol[j] >= 0 && coalesced) vload<VEC>(act[j], a.actions + (long long)ccol[j] * a.act_stride_col + env0);
        else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) act[j][i] = flex_action(a, ccol[j], env0 + i);
        }
        if (hdr[j] >= 0.0f) {
            const float* sp = f.ev_state + (long long)(int)hdr[j] * a.n_env + env0;
            vload<VEC>(soc[j], sp); vload<VEC>(ef[j], sp + ev_plane); vload<VEC>(deg[j], sp + 2 * ev_plane);
        }
    }
#pragma unroll
    for (int j = 0; j < CL_MAXW; ++j) {
        if (whdr[j] == CLWT_EMPTY) continue;
        vload<VEC>(winit[j], f.wm_state + (long long)(u * CL_MAXW + j) * a.n_env + env0);
        if (wcol[j] >= 0 && coalesced) vload<VEC>(wact[j], a.actions + (long long)wcol[j] * a.act_stride_col + env0);
        else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) wact[j][i] = flex_action(a, wcol[j], env0 + i);
        }
    }

    // ---- charging constraints (Building._apply_charging_constraints_to_actions, building.py:901-989): scale the positive
    //      requests action * max_charging_power under the building limit, then under every phase limit in order ----
    const uint32_t* __restrict__ cc = f.cons_params ? f.cons_params + (long long)u * CL_NCC : nullptr;
    if (cc && (cc[CLCC_FLAGS] & 1u)) {
        const float limit_b = cl::pw(cc, CLCC_BUILDING_LIMIT);
        float limit_p[CL_MAXPH], maxc[CL_MAXC];
        uint32_t mask_p[CL_MAXPH];
#pragma unroll
        for (int p = 0; p < CL_MAXPH; ++p) { limit_p[p] = cl::pw(cc, CLCC_PHASE_LIMIT0 + p); mask_p[p] = cc[CLCC_PHASE_MASK0 + p]; }
#pragma unroll
        for (int j = 0; j < CL_MAXC; ++j) maxc[j] = hdr[j] == CLCT_EMPTY ? 0.0f : cl::pw(cp0 + j * CL_NCP, CLC_MAX_CHARGE);
        const float dt = cl::pw(cp0, CLC_DT_HOURS);
        float viol[VEC], head_b[VEC], head_p[CL_MAXPH][VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float req[CL_MAXC], sc[CL_MAXC], total = 0.0f, v = 0.0f;
#pragma unroll
            for (int j = 0; j < CL_MAXC; ++j) {
                const bool pos = hdr[j] != CLCT_EMPTY && act[j][i] > 0.0f && maxc[j] > 0.0f;
                req[j] = pos ? act[j][i] * maxc[j] : 0.0f;
                sc[j] = 1.0f;
                total += req[j];
            }
            if (limit_b >= 0.0f && total > limit_b) {
                const float s = limit_b == 0.0f ? 0.0f : limit_b / total;
#pragma unroll
                for (int j = 0; j < CL_MAXC; ++j) sc[j] *= s;
                v += total - limit_b;
            }
#pragma unroll
            for (int p = 0; p < CL_MAXPH; ++p) {
                if (!(limit_p[p] >= 0.0f)) continue;
                float sum = 0.0f;
#pragma unroll
                for (int j = 0; j < CL_MAXC; ++j) sum += (mask_p[p] >> j & 1u) ? req[j] * sc[j] : 0.0f;
                if (sum > limit_p[p]) {
                    const float s = limit_p[p] == 0.0f ? 0.0f : limit_p[p] / sum;
#pragma unroll
                    for (int j = 0; j < CL_MAXC; ++j) sc[j] *= (mask_p[p] >> j & 1u) ? s : 1.0f;
                    v += sum - limit_p[p];
                }
            }
            float used = 0.0f;
#pragma unroll
            for (int j = 0; j < CL_MAXC; ++j) {
                const float scaled = req[j] * sc[j];
                used += scaled;
                if (hdr[j] != CLCT_EMPTY && act[j][i] > 0.0f)
                    act[j][i] = maxc[j] > 0.0f ? fmaxf(0.0f, fminf(act[j][i], scaled / maxc[j])) : 0.0f;
                req[j] = scaled;
            }
            viol[i] = v * dt;
            head_b[i] = limit_b - used;
#pragma unroll
            for (int p = 0; p < CL_MAXPH; ++p) {
                float sum = 0.0f;
#pragma unroll
                for (int j = 0; j < CL_MAXC; ++j) sum += (mask_p[p] >> j & 1u) ? req[j] : 0.0f;
                head_p[p][i] = limit_p[p] - sum;
            }
        }
        const long long fpc = (long long)f.n_flex_bldg * a.n_env, oc = (long long)u * a.n_env + env0;
        pstore<VEC, NT>(f.flex_out + CLX_VIOLATION * fpc + oc, viol);
        pstore<VEC, NT>(f.flex_out + CLX_HEADROOM * fpc + oc, head_b);
#pragma unroll
        for (int p = 0; p < CL_MAXPH; ++p) pstore<VEC, NT>(f.flex_out + (CLX_HEADROOM_PHASE0 + p) * fpc + oc, head_p[p]);
    }

    float chargers[VEC], wms[VEC], k0[VEC], kneg[VEC], kpos[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) chargers[i] = wms[i] = k0[i] = kneg[i] = kpos[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < CL_MAXC; ++j) {
        if (hdr[j] == CLCT_EMPTY) continue;
        const int c = u * CL_MAXC + j;
        const uint32_t* __restrict__ cp = cp0 + j * CL_NCP;
        const float* __restrict__ cr = cr0 + j * CL_NCF;
        const int k = (int)hdr[j];
        const float eff_c = cl::pw(cp, CLC_EFF), inv_eff_c = cl::pw(cp, CLC_INV_EFF), dt = cl::pw(cp, CLC_DT_HOURS);
        const bool curved = cp[CLC_CURVE_CHARGE_N] != 0u || cp[CLC_CURVE_DISCHARGE_N] != 0u;
        const float max_c = cl::pw(cp, CLC_MAX_CHARGE), min_c = cl::pw(cp, CLC_MIN_CHARGE);
        const float max_d = cl::pw(cp, CLC_MAX_DISCHARGE), min_d = cl::pw(cp, CLC_MIN_DISCHARGE);
        float energy[VEC], cons[VEC];
        // electric_vehicle_charger.py:306-322: requested energy, clamped to the charger's power range
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            energy[i] = act[j][i] > 0.0f ? fmaxf(fminf(act[j][i] * max_c * dt, max_c), min_c)
                      : act[j][i] < 0.0f ? fmaxf(fminf(act[j][i] * max_d * dt, -min_d), -max_d) : 0.0f;
            cons[i] = 0.0f;
        }
        if (k >= 0) {
            const uint32_t* __restrict__ ep = f.ev_params + (long long)k * CL_NP;
            const float* __restrict__ er = cr + CLCT_RULE_STEP - CLEV_RULE_STEP;      // the EV's rules, copied into the charger row
            float* sp = f.ev_state + (long long)k * a.n_env + env0;
            cl::BattP P;
            cl::load_batt(P, ep);
            const float cap = P.cap, min_cap = P.omd * cap, soc0 = cl::pw(ep, CLP_L_SOC0);
            const float required = cr[CLCT_REQUIRED_SOC], hours = cr[CLCT_DEPARTURE];
            const float reach_c = max_c * hours, reach_d = max_d * hours;
            bool charged = false;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float prev = soc[j][i];                       // soc[t - 1]; at t = 0 soc[0] itself (energy_model.py:661-666)
                float now = a.t == 0 ? prev : flex_begin_soc(a, er, row, k, env0 + i, prev);
                if (act[j][i] != 0.0f) {
                    cl::State S;
                    S.soc = prev; S.eff = ef[j][i]; S.degcap = deg[j][i]; S.cs = S.hs = S.ds = 0.0f;
                    float eff = eff_c, inv_eff = inv_eff_c;
                    if (curved) {                   // Charger.get_efficiency(|action|, charging)
                        const bool chg = act[j][i] > 0.0f;
                        const float on_curve = chg ? flex_curve(cp, CLC_CURVE_CHARGE_N, CLC_CURVE_CHARGE_X, CLC_CURVE_CHARGE_Y, fabsf(act[j][i]))
                                                   : flex_curve(cp, CLC_CURVE_DISCHARGE_N, CLC_CURVE_DISCHARGE_X, CLC_CURVE_DISCHARGE_Y, fabsf(act[j][i]));
                        const bool has = cp[chg ? CLC_CURVE_CHARGE_N : CLC_CURVE_DISCHARGE_N] != 0u;
                        eff = has ? on_curve : eff_c;
                        inv_eff = has ? 1.0f / on_curve : inv_eff_c;
                    }
                    const float to_battery = energy[i] * (act[j][i] > 0.0f ? eff : inv_eff);
                    const float eb = cl::battery_energy(P, to_battery * P.r, S);      // Battery.charge (energy_model.py:1027-1057)
                    now = S.soc; ef[j][i] = S.eff; deg[j][i] = S.degcap;
                    cons[i] = eb >= 0.0f ? eb * inv_eff : eb * eff;                     // electric_vehicle_charger.py:329
                    charged = true;
                }
                soc[j][i] = now;
                if (a.want_reward) {
                    // Electric_Vehicles_Reward_Function.calculate_ev_penalty, everything but the 1/(1+|MARL|) factor and the
                    // sign of the building net (reward_function.py:466-529)
                    const float soc_prev = a.t == 0 ? soc0 : prev;                      // building.py:1355
                    const float held = fmaf(soc_prev, cap, energy[i]);
                    if (held > cap || held < min_cap) k0[i] += f.weights[CLEW_BATTERY_LIMITS];
                    const float diff = now - required, diff_kwh = diff * cap;
                    if (diff_kwh > reach_c) k0[i] += f.weights[CLEW_SOC_IMPOSSIBLE];
                    if (hours == 0.0f) {
                        if (diff > -0.25f && diff <= -0.10f) k0[i] += 2.0f * f.weights[CLEW_SOC_UNDER];
                        else if (diff <= -0.25f) k0[i] += f.weights[CLEW_SOC_UNDER] * f.weights[CLEW_SOC_UNDER];
                        else if (diff > -0.10f && diff <= 0.10f) k0[i] += f.weights[CLEW_CLOSE_SOC];
                    }
                    if (fabsf(diff_kwh) <= fmaxf(reach_c, reach_d)) k0[i] += f.weights[CLEW_CLOSE_SOC] / (hours + 0.1f);
                    if (energy[i] > 0.0f) { kneg[i] += f.weights[CLEW_EXTRA_SELF_PRODUCTION]; kpos[i] += -0.5f * f.weights[CLEW_SELF_EV_CONSUMPTION]; }
                    else if (energy[i] < 0.0f) { kneg[i] += -0.5f * f.weights[CLEW_EXTRA_SELF_PRODUCTION]; kpos[i] += f.weights[CLEW_SELF_EV_CONSUMPTION]; }
                }
            }
            pstore<VEC, NT>(sp, soc[j]);
            if (charged) { pstore<VEC, NT>(sp + ev_plane, ef[j]); pstore<VEC, NT>(sp + 2 * ev_plane, deg[j]); }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) chargers[i] += cons[i];
        if (f.charger_out) {
            pstore<VEC, NT>(f.charger_out + (long long)c * a.n_env + env0, cons);
            pstore<VEC, NT>(f.charger_out + ((long long)f.n_flex_bldg * CL_MAXC + c) * a.n_env + env0, energy);
        }
    }
#pragma unroll
    for (int j = 0; j < CL_MAXW; ++j) {
        if (whdr[j] == CLWT_EMPTY) continue;
        const float* __restrict__ wr = wr0 + j * CL_NWF;
        const bool new_window = a.t > 0 && wr[CLWT_NEW_WINDOW] != 0.0f, open = whdr[j] != 0.0f;
        const float load = wr[CLWT_LOAD];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            bool initiated = winit[j][i] != 0.0f && !new_window;              // energy_model.py:1303-1312
            if (!initiated && wact[j][i] > 0.0f && open) {                     // energy_model.py:1320-1330
                initiated = true;
                wms[i] += load;
            }
            winit[j][i] = initiated ? 1.0f : 0.0f;
        }
        pstore<VEC, NT>(f.wm_state + (long long)(u * CL_MAXW + j) * a.n_env + env0, winit[j]);
    }
    if (!live) return;
    const long long fp = (long long)f.n_flex_bldg * a.n_env, o = (long long)u * a.n_env + env0;
    float total[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) total[i] = chargers[i] + wms[i];
    pstore<VEC, NT>(f.flex_out + CLX_LOAD * fp + o, total);
    if (a.want_chargers) pstore<VEC, NT>(f.flex_out + CLX_CHARGERS * fp + o, chargers);
    if (a.want_reward) {
        pstore<VEC, NT>(f.flex_out + CLX_RW_K0 * fp + o, k0);
        pstore<VEC, NT>(f.flex_out + CLX_RW_KNEG * fp + o, kneg);
        pstore<VEC, NT>(f.flex_out + CLX_RW_KPOS * fp + o, kpos);
    }
}

__global__ void cl_flex_reset_kernel(const cl_flex f, const int32_t* __restrict__ env_row0, int n_env) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n_ev_cells = (long long)f.n_ev * n_env, n_wm_cells = (long long)f.n_flex_bldg * CL_MAXW * n_env;
    if (i < n_ev_cells) {
        const int k = (int)(i / n_env), env = (int)(i - (long long)k * n_env);
        const int row = env_row0 ? env_row0[env / CL_ROW0_BLOCK] : 0;
        const uint32_t* ep = f.ev_params + (long long)k * CL_NP;
        const float rule = f.ev_ts[((long long)row * f.n_ev + k) * CL_NEVF + CLEV_RULE_RESET];
        f.ev_state[i] = rule >= 0.0f ? rule : cl::pw(ep, CLP_L_SOC0);
        f.ev_state[n_ev_cells + i] = cl::pw(ep, CLP_L_EFF0);
        f.ev_state[2 * n_ev_cells + i] = cl::pw(ep, CLP_L_CAP);
    }
    if (i < n_wm_cells) f.wm_state[i] = 0.0f;
}

}  // namespace
