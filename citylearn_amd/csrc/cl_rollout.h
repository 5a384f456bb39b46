// cl_rollout.h -- mode B: K consecutive environment steps in ONE launch with the per-unit state held in VGPRs.
// Included by cl_kernels.hip after the shared helpers (StepArgs, vload/vstore, district_reduce).
//
// What it replaces in the reference: K iterations of `Agent.learn`'s inner loop (agents/base.py:127-186) =
// `Agent.predict` (uniform `action_space.sample()`, agents/base.py:188-209) + `CityLearnEnv.step`
// (citylearn.py:978-1056), for every env of the batch.  Actions come either from an open-loop tensor
// (k, column, env) or from an on-device policy: a = low + u * (high - low), u = Philox4x32-10(seed; env, column, t).
//
// HBM traffic per unit is (state read + write) / K plus the last step's outputs -- the kernel is VALU-bound, not
// HBM-bound (mode B); the time-series rows of steps t0 .. t0+K-1 are scalar loads that hit in L2.
#pragma once

#include "cl_philox.h"

#ifdef __HIPCC__
namespace {

struct RolloutArgs {
    StepArgs s;                        // tables, state, (open-loop) actions, outputs, dims; s.t is unused
    long long act_stride_step;
    const float* __restrict__ act_low;
    const float* __restrict__ act_high;
    float* __restrict__ ret_env;       // [n_env] += sum over the K steps of the district reward (may be NULL)
    unsigned long long seed;
    int t0, k_steps;
};

// Electrical-storage action with the Philox block cached across four steps (wave-uniform refresh).
// (The block is kept as four separate words per env: held as one `U4` the compiler turned the wave-uniform word select into a
//  dynamically indexed array -- a 16-byte scratch store and a scratch load per env and step inside the rollout loop.)
struct PhiloxCache { uint32_t w0, w1, w2, w3; };
template <int VEC>
CL_DEV void rollout_action_cached(float (&dst)[VEC], PhiloxCache (&cache)[VEC], const RolloutArgs& r, int col, int env0, int t, int k, bool live);

// `live`: lanes past the end of the batch (ragged last tile) must not touch the open-loop action tensor.
template <int VEC>
CL_DEV void rollout_action(float (&dst)[VEC], const RolloutArgs& r, int col, int env0, int t, int k, bool live) {
    if (col < 0 || (r.s.actions && !live)) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) dst[i] = 0.0f;
        return;
    }
    if (r.s.actions) {
        const float* p = r.s.actions + (long long)k * r.act_stride_step + (long long)col * r.s.act_stride_col;
        if (r.s.act_stride_env == 1) vload<VEC>(dst, p + env0);
        else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) dst[i] = p[(long long)(env0 + i) * r.s.act_stride_env];
        }
    } else {
        const float lo = r.act_low[col], span = r.act_high[col] - lo;
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            dst[i] = fmaf(cl::philox_u01(r.seed, (uint32_t)(env0 + i) + r.s.env_offset, (uint32_t)col, (uint32_t)t), span, lo);
    }
}

template <int VEC>
CL_DEV void rollout_action_cached(float (&dst)[VEC], PhiloxCache (&cache)[VEC], const RolloutArgs& r, int col, int env0, int t, int k, bool live) {
    if (col < 0 || r.s.actions) { rollout_action<VEC>(dst, r, col, env0, t, k, live); return; }
    if (k == 0 || (t & 3) == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const cl::U4 blk = cl::philox_block(r.seed, (uint32_t)(env0 + i) + r.s.env_offset, (uint32_t)col, (uint32_t)t >> 2);
            cache[i].w0 = blk.w[0]; cache[i].w1 = blk.w[1]; cache[i].w2 = blk.w[2]; cache[i].w3 = blk.w[3];
        }
    }
    const float lo = r.act_low[col], span = r.act_high[col] - lo;
    const int sel = t & 3;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        // (words copied to locals first: a select between struct members becomes a select between their ADDRESSES, which parks the
        //  struct in scratch memory)
        const uint32_t w0 = cache[i].w0, w1 = cache[i].w1, w2 = cache[i].w2, w3 = cache[i].w3;
        const uint32_t lo_w = (sel & 1) ? w1 : w0, hi_w = (sel & 1) ? w3 : w2;
        const uint32_t word = (sel & 2) ? hi_w : lo_w;
        dst[i] = fmaf(cl::u01(word), span, lo);
    }
}

// MB = buildings owned by one wave (wave w owns w, w + nw, ...): their State stays in registers for all K steps.
// Districts beyond MB x 16 buildings (round 5; BASELINE config 4's 1024): the buildings are cut into chunks of `b_chunk` = MB x nw along
// gridDim.y exactly as the one-step launches cut them -- wave w of workgroup row y owns y b_chunk + w (+ nw) -- and a workgroup ends with
// the last step's chunk partial sums (and its chunk's share of the K-step return) in the scratch rows of out_bldg's reserved plane, which
// cl_finish_kernel folds once per LAUNCH, i.e. once per K steps.  Per unit and step the HBM traffic drops from 36 B (mode A) to 24 / K + 12 / K;
// what a chunked launch cannot do is a reward that couples the buildings inside a step (MARL: the host keeps cl_rollout_seq_f32 for it).
// PREC = 2: CLD_F64_CHAIN (cl::battery_charge_chain; the degraded-capacity plane carries the capacity loss)
template <int VEC, bool FULL, int MB, bool PIN = true, bool CHUNK = false, int PREC = 0>
__global__ void __launch_bounds__(1024) cl_rollout_kernel(const RolloutArgs r) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [nw][NQ][64*VEC]
    const StepArgs& a = r.s;
    constexpr int TILE = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int env0 = blockIdx.x * TILE + lane * VEC;
    const bool live = env0 < a.n_env;
    const long long plane = (long long)a.n_bldg * a.ld;            // (a.ld == a.n_env unless cl_dims.env_pitch pads the rows)
    const int rkind = (a.flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT;
    const bool quirk = a.flags & CLD_REF_T0_QUIRK;
    const bool detail = a.flags & CLD_WRITE_DETAIL;

    // (CHUNK is a template parameter: the scalar register file of these kernels is full -- numbered_sgpr = 100 with two dozen values parked in
    //  vector-register lanes -- and the workgroup-row index alone moved the one-row instantiations' allocation enough to reserve scratch memory)
    const int b_lo = CHUNK ? blockIdx.y * a.b_chunk : 0;
    const int b_hi = CHUNK ? min(a.n_bldg, b_lo + a.b_chunk) : a.n_bldg;
    // the table rows of this wave's first building: inside the K-step loop neither the chunk nor the wave index is needed for a row address
    // (the scalar register file of these kernels is full: one more value alive across the loop spills)
    const float* __restrict__ ts_w = a.ts + (long long)(b_lo + w) * CL_NF;
    cl::Bp B[MB];
    cl::State S[MB][VEC];
    bool own[MB];
    long long off[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int b = b_lo + w + m * a.nw;
        own[m] = b < b_hi;
        const int bc = own[m] ? b : (CHUNK ? min(b_lo + w, a.n_bldg - 1) : w);
        off[m] = (long long)bc * a.ld + env0;
        cl::load_bp<FULL>(B[m], a.params + (long long)bc * CL_NP);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            S[m][i] = {0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if (live && own[m]) {
                const long long o = off[m] + i;
                S[m][i].soc = a.state[CLS_B_SOC * plane + o];
                S[m][i].eff = a.state[CLS_B_EFF * plane + o];
                S[m][i].degcap = a.state[CLS_B_DEGCAP * plane + o];
                if constexpr (FULL) {
                    S[m][i].cs = a.state[CLS_CS_SOC * plane + o];
                    S[m][i].hs = a.state[CLS_HS_SOC * plane + o];
                    S[m][i].ds = a.state[CLS_DS_SOC * plane + o];
                }
            }
        }
    }
    [[maybe_unused]] cl::BattP Bv[MB];
    if constexpr (!FULL) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            Bv[m] = B[m].batt;
            // PIN: the curve parameters in VGPRs for all K steps.  Not for one env per lane x two buildings per wave while the whole launch
            // is resident at once (17 x 32 768: 4.5 waves per SIMD): 24 more VGPRs (67 -> 90) leave five waves per SIMD, the second 9-wave
            // workgroup of a CU then only fits where the dispatcher happens to start it, and every launch pays ~3 us (2.77 vs 2.60 us per
            // step at K = 24); from 65 536 envs up the launch runs in several generations anyway and the pins win (4.99 vs 5.75 us per step).
            if constexpr (!PIN) continue;
            CL_PIN_V(Bv[m].cpc_a0); CL_PIN_V(Bv[m].cpc_b0); CL_PIN_V(Bv[m].cpc_a1); CL_PIN_V(Bv[m].cpc_b1);
            CL_PIN_V(Bv[m].pec_a0); CL_PIN_V(Bv[m].pec_b0); CL_PIN_V(Bv[m].pec_a1); CL_PIN_V(Bv[m].pec_b1);
            CL_PIN_V(Bv[m].pec_a2); CL_PIN_V(Bv[m].pec_b2); CL_PIN_V(Bv[m].pec_a3); CL_PIN_V(Bv[m].pec_b3);
        }
    }
    float ret[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) ret[i] = 0.0f;
    float q_net[VEC], q_cost[VEC], q_em[VEC], q_rw[VEC];
    cl::Out last[MB][VEC];
    float last_rw[MB][VEC];
    PhiloxCache rnd[MB][VEC];

    const int row0 = a.env_row0 ? a.env_row0[(blockIdx.x * 64 * VEC) / CL_ROW0_BLOCK] : 0;   // workgroup-uniform
    for (int k = 0; k < r.k_steps; ++k) {
        const int t = r.t0 + k;
#pragma unroll
        for (int i = 0; i < VEC; ++i) q_net[i] = q_cost[i] = q_em[i] = q_rw[i] = 0.0f;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (!own[m]) continue;                                       // wave-uniform
            cl::Row R;
            cl::load_row<FULL>(R, ts_w + ((long long)(t + row0) * a.n_bldg + m * a.nw) * CL_NF, B[m].flags,
                               FULL ? ts_w + ((long long)(a.n_steps - 1 + row0) * a.n_bldg + m * a.nw) * CL_NF : nullptr);
            float a_es[VEC], a_cs[VEC], a_hs[VEC], a_ds[VEC], a_cd[VEC], a_hd[VEC];
            rollout_action_cached<VEC>(a_es, rnd[m], r, B[m].a_es, env0, t, k, live);
            if constexpr (FULL) {
                rollout_action<VEC>(a_cs, r, B[m].a_cs, env0, t, k, live);
                rollout_action<VEC>(a_hs, r, B[m].a_hs, env0, t, k, live);
                rollout_action<VEC>(a_ds, r, B[m].a_ds, env0, t, k, live);
                if (B[m].a_coh >= 0) {
                    float c[VEC];
                    rollout_action<VEC>(c, r, B[m].a_coh, env0, t, k, live);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) { a_cd[i] = fabsf(fminf(c[i], 0.0f)); a_hd[i] = fabsf(fmaxf(c[i], 0.0f)); }
                } else {
                    rollout_action<VEC>(a_cd, r, B[m].a_cd, env0, t, k, live);
                    rollout_action<VEC>(a_hd, r, B[m].a_hd, env0, t, k, live);
                }
            }
            if constexpr (!FULL) {
                // the lean unit with its per-building uniforms outside the env loop (same expressions as cl::unit_step<false> /
                // cl::unit_reward<false>: see cl_step_lean_kernel); the battery's curve parameters sit in VGPRs for all K steps
                const bool first = quirk && t == 0;
                float c_ns = first ? 3.0f * R.nsl : R.nsl, sol = R.sol;
                const float cbk = first ? 2.0f : 1.0f;
                if constexpr (VEC > 1) { CL_PIN_V(c_ns); CL_PIN_V(sol); }
                const bool batt = B[m].flags & CLF_BATTERY;
                float nets[VEC], socs[VEC], rws[VEC];
                [[maybe_unused]] cl::BattC bc;
                if constexpr (PREC == 2) {
                    if (batt) cl::load_battc(bc, B[m].p);          // (scalar loads every step: 19 doubles per building do not fit beside the fp32 block)
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float eb = 0.0f;
                    if constexpr (PREC == 2) {
                        if (batt) eb = cl::battery_charge_chain(bc, a_es[i], INFINITY, S[m][i]);
                    } else if (batt) eb = cl::battery_energy(Bv[m], a_es[i] * Bv[m].pdt, S[m][i]);
                    nets[i] = fmaf(c_ns + cbk * eb, B[m].r, sol);
                    socs[i] = S[m][i].soc;
                }
                cl::lean_rewards<VEC>(rkind, B[m], socs, nets, rws);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    last[m][i].net = nets[i]; last_rw[m][i] = rws[i];
                    q_net[i] += nets[i]; q_cost[i] += cl::mul_rn(nets[i], R.price); q_em[i] += fmaxf(0.0f, nets[i] * R.carbon); q_rw[i] += rws[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const cl::Act act = {a_cs[i], a_hs[i], a_ds[i], a_es[i], a_cd[i], a_hd[i]};
                    cl::unit_step<FULL, PREC>(B[m], R, t, quirk, act, S[m][i], last[m][i]);
                    const float rw = cl::unit_reward<FULL>(rkind, B[m], S[m][i], last[m][i].net);
                    last_rw[m][i] = rw;
                    q_net[i] += last[m][i].net; q_cost[i] += last[m][i].cost; q_em[i] += last[m][i].emission; q_rw[i] += rw;
                }
            }
        }
        if (rkind == CLR_MARL) {
            // the MARL reward couples the buildings through the district net of THIS step: one LDS exchange per step
            vstore<VEC>(lds + (size_t)w * TILE + lane * VEC, q_net);
            __syncthreads();
            float dnet[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) dnet[i] = 0.0f;
            for (int kk = 0; kk < a.nw; ++kk) {
                float part[VEC];
                vload<VEC>(part, lds + (size_t)kk * TILE + lane * VEC);
#pragma unroll
                for (int i = 0; i < VEC; ++i) dnet[i] += part[i];
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (!own[m]) continue;
#pragma unroll
                for (int i = 0; i < VEC; ++i) { last_rw[m][i] = cl::marl_reward(last[m][i].net, dnet[i]); ret[i] += last_rw[m][i]; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) ret[i] += q_rw[i];
        }
    }

    // ---- write back: carried state, the last step's per-building outputs, district sums, episode-return partials ----
    if (live) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (!own[m]) continue;
            float v[VEC];
#define CL_PUT(base, plane_id, expr)                                   \
    _Pragma("unroll") for (int i = 0; i < VEC; ++i) v[i] = (expr);      \
    vstore<VEC>(base + (plane_id) * plane + off[m], v);
            if (B[m].flags & CLF_BATTERY) {
                CL_PUT(a.state, CLS_B_SOC, S[m][i].soc) CL_PUT(a.state, CLS_B_EFF, S[m][i].eff) CL_PUT(a.state, CLS_B_DEGCAP, S[m][i].degcap)
            }
            if constexpr (FULL) {
                if (B[m].flags & CLF_COOL_STO) { CL_PUT(a.state, CLS_CS_SOC, S[m][i].cs) }
                if (B[m].flags & CLF_HEAT_STO) { CL_PUT(a.state, CLS_HS_SOC, S[m][i].hs) }
                if (B[m].flags & CLF_DHW_STO) { CL_PUT(a.state, CLS_DS_SOC, S[m][i].ds) }
            }
            if (r.k_steps > 0) {
                CL_PUT(a.out_bldg, CLO_NET, last[m][i].net)
                CL_PUT(a.out_bldg, CLO_REWARD, last_rw[m][i])
                if (FULL && detail) {
                    CL_PUT(a.out_bldg, CLO_B_EB, last[m][i].eb) CL_PUT(a.out_bldg, CLO_COOL_DEM, last[m][i].cool_dem)
                    CL_PUT(a.out_bldg, CLO_C_COOL, last[m][i].c_cool) CL_PUT(a.out_bldg, CLO_C_HEAT, last[m][i].c_heat)
                    CL_PUT(a.out_bldg, CLO_C_DHW, last[m][i].c_dhw) CL_PUT(a.out_bldg, CLO_C_NSL, last[m][i].c_ns)
                    CL_PUT(a.out_bldg, CLO_BASE_NET, last[m][i].base_net) CL_PUT(a.out_bldg, CLO_EXPECTED, last[m][i].expected)
                    CL_PUT(a.out_bldg, CLO_SERVED, last[m][i].served) CL_PUT(a.out_bldg, CLO_NET_WS, last[m][i].net_ws)
                    CL_PUT(a.out_bldg, CLO_HEAT_DEM, last[m][i].heat_dem) CL_PUT(a.out_bldg, CLO_DHW_DEM, last[m][i].dhw_dem)
                    CL_PUT(a.out_bldg, CLO_SE_COOL, last[m][i].se_cool) CL_PUT(a.out_bldg, CLO_SE_HEAT, last[m][i].se_heat)
                    CL_PUT(a.out_bldg, CLO_SE_DHW, last[m][i].se_dhw)
                }
            }
#undef CL_PUT
        }
    }
    if (r.k_steps > 0) {
        if (rkind == CLR_MARL) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                q_rw[i] = 0.0f;
#pragma unroll
                for (int m = 0; m < MB; ++m) q_rw[i] += own[m] ? last_rw[m][i] : 0.0f;
            }
        }
        // district sums of the last step (for MARL the reward plane / sum were finished above: pass kind DEFAULT)
        district_reduce<VEC>(a, lds, w, lane, env0, live, plane, rkind == CLR_MARL ? (int)CLR_DEFAULT : rkind, q_net, q_cost, q_em,
                             q_rw, a.nw);
    }
    if (r.ret_env) {
        __syncthreads();
        vstore<VEC>(lds + (size_t)w * TILE + lane * VEC, ret);
        __syncthreads();
        const int tile_env0 = blockIdx.x * TILE;
        // (chunked: this workgroup row's share of the return goes to its scratch row behind the n_chunks x NQ partial sums; cl_finish_kernel adds the rows)
        for (int e = threadIdx.x; e < TILE; e += blockDim.x) {
            float s = 0.0f;
            for (int kk = 0; kk < a.nw; ++kk) s += lds[(size_t)kk * TILE + e];
            if (tile_env0 + e < a.n_env) {
                if constexpr (CHUNK) a.out_bldg[(long long)CLO_RESERVED * plane + ((long long)a.n_chunks * NQ + blockIdx.y) * a.n_env + tile_env0 + e] = s;
                else r.ret_env[tile_env0 + e] += s;
            }
        }
    }
}

#ifndef CL_TU_NOSLP
// ---- mode B for thermal / outage districts around the PACK-GENERIC unit of cl_full.h (round 6) ------------------------------------------------
// cl_rollout_kernel<1, true, 1> above steps a thermal building with the scalar unit of cl_unit.h at one env per lane: 384 vector instructions per
// unit-step, which made the fused rollout of the 1024-building thermal district SLOWER per step than mode A (17.6 vs 13.2 us at 1024 x 1024, fp32).
// Here the same K-step loop runs clv::unit_step -- the arithmetic of the thermal step kernels, two envs per lane in packed fp32 where the battery map
// is fp32 (the float64 chain keeps one env per lane, like every thermal kernel) -- with the unit state in registers for all K steps, one building
// per wave, chunks of up to 16 buildings along gridDim.y.  Same action streams, same district reduction, same return rows as cl_rollout_kernel.
// No detail planes (the host keeps cl_rollout_kernel for CLD_WRITE_DETAIL).  Lives in the main translation unit: it WANTS the packed instructions.
template <int VEC>
CL_DEV typename Vec<VEC>::type rollout_action_f(const RolloutArgs& r, int col, int env0, int t, int k, bool live) {
    float d[VEC];
    rollout_action<VEC>(d, r, col, env0, t, k, live);
    if constexpr (VEC == 1) return d[0];
    else { typename Vec<VEC>::type v; _Pragma("unroll") for (int i = 0; i < VEC; ++i) v[i] = d[i]; return v; }
}

// The on-device policy of a thermal building draws up to six columns per step, and a Philox block is 4 words = the column's next FOUR steps (cl_philox.h):
// drawn per step and column, the blocks were two thirds of this kernel's vector work (2 x 125 of ~ 600 instructions per step at two envs per lane,
// 32 of them quarter-rate 32 x 32 multiplies).  Each wave keeps the blocks of its building's first CL_ROLLOUT_SLOTS<VEC> active columns in LDS --
// [slot][word][64 VEC], refreshed where t % 4 == 0 (and at the launch's first step), one ds_read per draw; the registers do not hold them (48 at
// two envs per lane, 105 in use).  A fifth / sixth column at two envs per lane has a one-word slot, redrawn every step.  ONE copy of the block
// function, in a loop that is not unrolled: six inlined copies (one per column, each behind its own wave-uniform branch) were scheduled into each
// other -- 128 registers and 112 bytes of scratch.  The region sits behind MARL's exchange row and aliases the district reduction's rows (the K loop
// ends with a barrier before anybody reduces).
template <int VEC> constexpr int CL_ROLLOUT_SLOTS = VEC == 1 ? 6 : 4;

template <int VEC> constexpr int CL_ROLLOUT_RND_ROWS = CL_ROLLOUT_SLOTS<VEC> * 4 + (6 - CL_ROLLOUT_SLOTS<VEC>);      // rows of 64 VEC words per wave

template <int VEC>
CL_DEV int rollout_rnd_row(int slot, int t) {
    return slot < CL_ROLLOUT_SLOTS<VEC> ? slot * 4 + (t & 3) : CL_ROLLOUT_SLOTS<VEC> * 3 + slot;
}

// (re)draw what step t needs.  `cols_lo` / `cols_hi`: the unit's active column ids in slot order, 16 bits each (host: n_act_cols <= 65 536);
// the loop walks the wide slots where t % 4 == 0 (and at the launch's first step) and the one-word slots every step -- nothing on three steps of
// four for a building of up to CL_ROLLOUT_SLOTS columns.  (A loop over the unit's seven candidate columns with its selects and skips was ~ 100
// scalar instructions and ~ 20 branches per step and wave: 4 waves per SIMD do not hide that -- SALU +77 %, vector-ALU busy 85 -> 66 %.)
template <int VEC>
CL_DEV void rollout_rnd_refresh(const RolloutArgs& r, float* __restrict__ rnd, unsigned long long cols_lo, uint32_t cols_hi, int n_slot, int env0, int lane, int t, int k) {
    constexpr int TILE = 64 * VEC;
    const bool four = k == 0 || (t & 3) == 0;
#pragma unroll 1
    for (int slot = four ? 0 : CL_ROLLOUT_SLOTS<VEC>; slot < n_slot; ++slot) {
        const uint32_t col = slot < 4 ? (uint32_t)(cols_lo >> (16 * slot)) & 0xffffu : (cols_hi >> (16 * (slot - 4))) & 0xffffu;
        cl::U4 blk[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) blk[i] = cl::philox_block(r.seed, (uint32_t)(env0 + i) + r.s.env_offset, col, (uint32_t)t >> 2);
        // what the rows hold is the ACTION, a = low + u (high - low): the column's bounds are read here, once per block, not once per draw
        // (two scalar loads and their round trip per column and step, from addresses that sat in spilled SGPRs)
        const float lo = r.act_low[col], span = r.act_high[col] - lo;
        if (slot < CL_ROLLOUT_SLOTS<VEC>) {
            float* c = rnd + (size_t)slot * 4 * TILE + lane * VEC;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) c[w * TILE + i] = fmaf(cl::u01(blk[i].w[w]), span, lo);
            }
        } else {
            float* c = rnd + (size_t)rollout_rnd_row<VEC>(slot, t) * TILE + lane * VEC;
#pragma unroll
            for (int i = 0; i < VEC; ++i) c[i] = fmaf(cl::u01(cl::philox_word(blk[i], (uint32_t)t & 3u)), span, lo);
        }
    }
}

// one column of the on-device policy from the wave's cached words / of the open-loop action tensor
template <int VEC>
CL_DEV typename Vec<VEC>::type rollout_action_drawn(const float* rnd, int slot, int col, int lane, int t) {
    constexpr int TILE = 64 * VEC;
    float d[VEC];
    if (col < 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) d[i] = 0.0f;
    } else {
        const float* w = rnd + (size_t)rollout_rnd_row<VEC>(slot, t) * TILE + lane * VEC;
#pragma unroll
        for (int i = 0; i < VEC; ++i) d[i] = w[i];
    }
    if constexpr (VEC == 1) return d[0];
    else { typename Vec<VEC>::type v; _Pragma("unroll") for (int i = 0; i < VEC; ++i) v[i] = d[i]; return v; }
}

template <int VEC>
CL_DEV typename Vec<VEC>::type rollout_action_open(const RolloutArgs& r, int col, int env0, int k, bool live) {
    float d[VEC];
    if (col < 0 || !live) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) d[i] = 0.0f;
    } else {
        const float* p = r.s.actions + (long long)k * r.act_stride_step + (long long)col * r.s.act_stride_col;
        if (r.s.act_stride_env == 1) vload<VEC>(d, p + env0);
        else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) d[i] = p[(long long)(env0 + i) * r.s.act_stride_env];
        }
    }
    if constexpr (VEC == 1) return d[0];
    else { typename Vec<VEC>::type v; _Pragma("unroll") for (int i = 0; i < VEC; ++i) v[i] = d[i]; return v; }
}

// MARL: the reward couples the buildings through the district net of the SAME step -- one LDS exchange (two barriers) per step, its own instantiation:
// with a barrier inside the K loop the no-clobber walk stops at it and calls every global read of the loop clobbered by the loop's LDS stores --
// the wave-uniform parameter reads came back as vector loads + v_readfirstlane (never chunked: host).
template <int VEC, bool CHUNK, int PREC, bool MARL = false>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4))) cl_rollout_full_kernel(const RolloutArgs r) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // max([nw][NQ][64*VEC], [nw][64*VEC] + [nw][slots][4][64*VEC])
    using F = typename Vec<VEC>::type;
    const StepArgs& a = r.s;
    constexpr int TILE = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int env0 = blockIdx.x * TILE + lane * VEC;
    const bool live = env0 < a.n_env;
    const long long plane = (long long)a.n_bldg * a.n_env;
    const int rkind = (a.flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT;
    const bool quirk = a.flags & CLD_REF_T0_QUIRK;
    const int b_lo = CHUNK ? blockIdx.y * a.b_chunk : 0;
    const int b_hi = CHUNK ? min(a.n_bldg, b_lo + a.b_chunk) : a.n_bldg;
    const int b = b_lo + w;
    const bool own = b < b_hi;                                       // wave-uniform
    const int bc = own ? b : min(b_lo, a.n_bldg - 1);
    const uint32_t* __restrict__ f = a.params + (long long)bc * CL_NP + CLP_F_FIRST;
    [[maybe_unused]] const uint32_t* __restrict__ grow = PREC == 2 ? a.params + (long long)bc * CL_NP : nullptr;
    const uint32_t flags = clv::uword<false>(f, 0);
    const int c_cs = (int)clv::uword<false>(f, 1), c_hs = (int)clv::uword<false>(f, 2), c_ds = (int)clv::uword<false>(f, 3), c_es = (int)clv::uword<false>(f, 4),
              c_cd = (int)clv::uword<false>(f, 5), c_hd = (int)clv::uword<false>(f, 6), c_coh = (int)clv::uword<false>(f, 7);
    const long long off = (long long)bc * a.n_env + env0;
    const F zero = (F)(0.0f), one = (F)(1.0f);
    clv::St<F> S = {zero, one, zero, zero, zero, zero};
    if (live && own) {
        if (flags & CLF_BATTERY) {
            S.soc = full_load<VEC>(a.state + CLS_B_SOC * plane + off); S.eff = full_load<VEC>(a.state + CLS_B_EFF * plane + off);
            S.degcap = full_load<VEC>(a.state + CLS_B_DEGCAP * plane + off);
        }
        if (flags & CLF_COOL_STO) S.cs = full_load<VEC>(a.state + CLS_CS_SOC * plane + off);
        if (flags & CLF_HEAT_STO) S.hs = full_load<VEC>(a.state + CLS_HS_SOC * plane + off);
        if (flags & CLF_DHW_STO) S.ds = full_load<VEC>(a.state + CLS_DS_SOC * plane + off);
    }
    float ret[VEC], q_net[VEC], q_cost[VEC], q_em[VEC], q_rw[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) ret[i] = 0.0f;
    F last_net = zero, last_rw = zero;
    float* rnd = lds + (size_t)a.nw * TILE + (size_t)w * CL_ROLLOUT_RND_ROWS<VEC> * TILE;      // this wave's drawn actions
    // slots in column order of the unit (wave-uniform), their column ids packed 16 bits each
    int n_slot = 0;
    unsigned long long cols_lo = 0ull;
    uint32_t cols_hi = 0u;
    auto slot_of = [&](int c) {
        if (c < 0) return 0;
        if (n_slot < 4) cols_lo |= (unsigned long long)(uint32_t)c << (16 * n_slot); else cols_hi |= (uint32_t)c << (16 * (n_slot - 4));
        return n_slot++;
    };
    const int s_es = slot_of(c_es), s_cs = slot_of(c_cs), s_hs = slot_of(c_hs), s_ds = slot_of(c_ds), s_coh = slot_of(c_coh),
              // (a combined cooling-or-heating column takes the place of the two device columns, building.py:1557-1564: at most six columns, six slots)
              s_cd = c_coh >= 0 ? 0 : slot_of(c_cd), s_hd = c_coh >= 0 ? 0 : slot_of(c_hd);
    const int row0 = a.env_row0 ? a.env_row0[(blockIdx.x * TILE) / CL_ROW0_BLOCK] : 0;   // workgroup-uniform
    for (int k = 0; k < r.k_steps; ++k) {
        const int t = r.t0 + k;
#pragma unroll
        for (int i = 0; i < VEC; ++i) q_net[i] = q_cost[i] = q_em[i] = q_rw[i] = 0.0f;
        if (own) {
            // The building's parameter block is RE-READ every step through the scalar cache: hoisted out of the loop (it is loop-invariant) its ~ 60
            // words outlive the SGPR file -- 93 v_writelane before the loop, 13 - 16 v_readlane per use inside it.  `z` is a zero the compiler cannot
            // see through.  (Scalar only while the K loop holds no barrier: MARL's note above the kernel.)
#ifndef CL_ROLLOUT_HOIST
#define CL_ROLLOUT_HOIST 0
#endif
            int z;
            if constexpr (CL_ROLLOUT_HOIST == 2) z = 0;
            else asm("s_mov_b32 %0, 0" : "=s"(z) : "s"(k));      // (not volatile: that would be a store to everything, as far as the no-clobber walk knows)
            const uint32_t* __restrict__ fz = f + z;
            [[maybe_unused]] const uint32_t* __restrict__ gz = PREC == 2 ? grow + z : nullptr;
            clv::FP B;
            clv::load_fp<false>(B, CL_ROLLOUT_HOIST == 1 ? f : fz);
            B.f = fz;
            cl::Row R;
            cl::load_row_scalar<true>(R, a.ts + ((long long)(t + row0) * a.n_bldg + bc) * CL_NF, B.flags, nullptr);
            clv::Ac<F> act;
            F a_coh = zero;
            if (r.s.actions) {
                act.es = rollout_action_open<VEC>(r, c_es, env0, k, live); act.cs = rollout_action_open<VEC>(r, c_cs, env0, k, live);
                act.hs = rollout_action_open<VEC>(r, c_hs, env0, k, live); act.ds = rollout_action_open<VEC>(r, c_ds, env0, k, live);
                act.cd = rollout_action_open<VEC>(r, c_cd, env0, k, live); act.hd = rollout_action_open<VEC>(r, c_hd, env0, k, live);
                a_coh = rollout_action_open<VEC>(r, c_coh, env0, k, live);
            } else {
                rollout_rnd_refresh<VEC>(r, rnd, cols_lo, cols_hi, n_slot, env0, lane, t, k);
                act.es = rollout_action_drawn<VEC>(rnd, s_es, c_es, lane, t); act.cs = rollout_action_drawn<VEC>(rnd, s_cs, c_cs, lane, t);
                act.hs = rollout_action_drawn<VEC>(rnd, s_hs, c_hs, lane, t); act.ds = rollout_action_drawn<VEC>(rnd, s_ds, c_ds, lane, t);
                act.cd = rollout_action_drawn<VEC>(rnd, s_cd, c_cd, lane, t); act.hd = rollout_action_drawn<VEC>(rnd, s_hd, c_hd, lane, t);
                a_coh = rollout_action_drawn<VEC>(rnd, s_coh, c_coh, lane, t);
            }
            if (c_coh >= 0) { act.cd = clv::vabs(clv::vmin(a_coh, zero)); act.hd = clv::vabs(clv::vmax(a_coh, zero)); }
            clv::Ou<F> O;
            const bool first = quirk && t == 0;
            if (R.outage) clv::unit_step<F, true, false, PREC>(B, R, t, first, act, S, O, gz);
            else clv::unit_step<F, false, false, PREC>(B, R, t, first, act, S, O, gz);
            const F rw = clv::unit_reward<F>(rkind, B, S, O.net);
            last_net = O.net; last_rw = rw;
            full_accumulate<VEC>(q_net, O.net); full_accumulate<VEC>(q_cost, O.cost); full_accumulate<VEC>(q_em, O.emission); full_accumulate<VEC>(q_rw, rw);
        }
        if constexpr (MARL) {
            // (never chunked: host) the MARL reward couples the buildings through the district net of THIS step: one LDS exchange per step
            vstore<VEC>(lds + (size_t)w * TILE + lane * VEC, q_net);
            __syncthreads();
            float dnet[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) dnet[i] = 0.0f;
            for (int kk = 0; kk < a.nw; ++kk) {
                float part[VEC];
                vload<VEC>(part, lds + (size_t)kk * TILE + lane * VEC);
#pragma unroll
                for (int i = 0; i < VEC; ++i) dnet[i] += part[i];
            }
            __syncthreads();
            if (own) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float n_i, rw_i;
                    if constexpr (VEC == 1) n_i = last_net; else n_i = last_net[i];
                    rw_i = cl::marl_reward(n_i, dnet[i]);
                    if constexpr (VEC == 1) last_rw = rw_i; else last_rw[i] = rw_i;
                    ret[i] += rw_i;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) ret[i] += q_rw[i];
        }
    }
    __syncthreads();                                                 // (the Philox blocks alias the reduction rows)
    if (live && own) {
        if (flags & CLF_BATTERY) {
            full_store<VEC, false>(a.state + CLS_B_SOC * plane + off, S.soc); full_store<VEC, false>(a.state + CLS_B_EFF * plane + off, S.eff);
            full_store<VEC, false>(a.state + CLS_B_DEGCAP * plane + off, S.degcap);
        }
        if (flags & CLF_COOL_STO) full_store<VEC, false>(a.state + CLS_CS_SOC * plane + off, S.cs);
        if (flags & CLF_HEAT_STO) full_store<VEC, false>(a.state + CLS_HS_SOC * plane + off, S.hs);
        if (flags & CLF_DHW_STO) full_store<VEC, false>(a.state + CLS_DS_SOC * plane + off, S.ds);
        if (r.k_steps > 0) {
            full_store<VEC, false>(a.out_bldg + CLO_NET * plane + off, last_net);
            full_store<VEC, false>(a.out_bldg + CLO_REWARD * plane + off, last_rw);
        }
    }
    if (r.k_steps > 0) {
        if constexpr (MARL) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float rw_i;
                if constexpr (VEC == 1) rw_i = last_rw; else rw_i = last_rw[i];
                q_rw[i] = own ? rw_i : 0.0f;
            }
        }
        district_reduce<VEC>(a, lds, w, lane, env0, live, plane, MARL ? (int)CLR_DEFAULT : rkind, q_net, q_cost, q_em, q_rw, a.nw);
    }
    if (r.ret_env) {
        __syncthreads();
        vstore<VEC>(lds + (size_t)w * TILE + lane * VEC, ret);
        __syncthreads();
        const int tile_env0 = blockIdx.x * TILE;
        for (int e = threadIdx.x; e < TILE; e += blockDim.x) {
            float s = 0.0f;
            for (int kk = 0; kk < a.nw; ++kk) s += lds[(size_t)kk * TILE + e];
            if (tile_env0 + e < a.n_env) {
                if constexpr (CHUNK) a.out_bldg[(long long)CLO_RESERVED * plane + ((long long)a.n_chunks * NQ + blockIdx.y) * a.n_env + tile_env0 + e] = s;
                else r.ret_env[tile_env0 + e] += s;
            }
        }
    }
}
#endif  // CL_TU_NOSLP

}  // namespace
#endif  // __HIPCC__
