// cl_kernels.hip -- HIP kernels (gfx950 / CDNA4) + the extern "C" boundary declared in include/citylearn_amd.h.
//
// Work decomposition ("2-D tile"): a workgroup owns ENV_TILE = 64*VEC consecutive envs and ALL buildings of
// the district.  Wave w of the workgroup advances buildings w, w+NW, w+2NW, ... so that the building index is
// wave-uniform: its parameter block and its time-series row are fetched with scalar loads into SGPRs, and the
// 64 lanes of the wave touch 64*VEC consecutive floats of every state / action / output plane (one fully
// coalesced request per plane).  District sums over buildings (net, cost, emission, reward) are reduced
// through LDS in a fixed order (wave-partials -> serial sum over waves), so results are bit-reproducible
// run to run -- no atomics on the step path.
#include "cl_trace.h"
#include "cl_unit.h"
#ifndef CL_SWAP_GRID
#define CL_SWAP_GRID false      // chunk-major grids for the always-chunked kernels (district_reduce's SWAP note): measured, NOT the default; -DCL_SWAP_GRID=true: the A/B build
#endif
#ifndef CL_LEAN_NT_LOADS
#define CL_LEAN_NT_LOADS true          // A/B build flag: false = no non-temporal hint on any plane load of the latency-ordered lean kernels (lean_step_body's note)
#endif
#include "cl_philox.h"

#include <type_traits>
#include <mutex>
#include <stdarg.h>
#include <string.h>
#include <stdio.h>

namespace {

thread_local char g_err[512] = {0};

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(CL_EHIP, "%s: %s", what, hipGetErrorString(e));
}

// Opt a kernel into more than the default 64 KB of dynamic LDS, ONCE per (kernel, device, size): hipFuncSetAttribute is a driver call, and
// the launches that need it (building-chunked 1024-building districts) sit on the per-step path (round-5 advisor finding).  The table only
// remembers what the driver has already been told -- idempotent, so it is not state a caller could observe.
hipError_t ensure_dynamic_lds(const void* fn, size_t bytes) {
    struct Seen { const void* fn; int dev; size_t bytes; };
    static Seen seen[64];
    static int n_seen = 0;
    static std::mutex mu;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < n_seen; ++i)
        if (seen[i].fn == fn && seen[i].dev == dev && seen[i].bytes >= bytes) return hipSuccess;
    if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); e != hipSuccess) return e;
    for (int i = 0; i < n_seen; ++i)
        if (seen[i].fn == fn && seen[i].dev == dev) { seen[i].bytes = bytes; return hipSuccess; }
    if (n_seen < 64) seen[n_seen++] = {fn, dev, bytes};
    return hipSuccess;
}

struct StepArgs {
    const uint32_t* __restrict__ params;
    const float* __restrict__ ts;
    float* __restrict__ state;
    const float* __restrict__ actions;
    float* __restrict__ out_bldg;
    float* __restrict__ out_env;
    float* __restrict__ kpi_bldg;
    float* __restrict__ kpi_env;
    long long act_stride_col, act_stride_env;
    const int32_t* __restrict__ env_row0;   // per-env-block episode offsets (cl_dims.env_row0) or null
    const float* __restrict__ flex_out;     // cl_flex.flex_out planes [CL_NX][n_flex_bldg][n_env] or null (cl_flex.h)
    int n_flex_bldg;
    unsigned env_offset;                    // cl_dims.env_offset (low 32 bits: the Philox counter word)
    float ev_penalty_coef;                  // > 0: some building has charging constraints; CLR_EV subtracts coef * violation
    int n_env, n_bldg, n_steps;
    int ld;        // floats between consecutive building rows of the state / out_bldg planes (cl_dims.env_pitch; = n_env unless padded).  (In the slot of
                   // the action-column count, which no kernel reads: one more word in this struct cost every detail kernel a scratch reservation)
    uint32_t flags;
    int t;
    int nw;        // waves per workgroup == building lanes
    int b_chunk;   // buildings per workgroup row (gridDim.y = n_chunks rows); == n_bldg when the grid is 1-D
    int n_chunks;
    int nt;        // plane stores carry the non-temporal hint (see pstore)
    int fused_finish;   // building-chunked launches: the last chunk of an env tile folds the chunk partial sums itself (district_reduce)
};

constexpr int CL_OBS_FUSED_BLDG = 32;          // buildings a fused observation list can address (= the lean kernel's 2 x 16)
constexpr int CL_OBS_FUSED_PER_BLDG = 4;      // observation columns per building the step launch writes itself

// largest launch (env x building units) whose plane stores carry the non-temporal hint (see pstore)
constexpr long long CL_NT_MAX_UNITS = 3ll << 20;
constexpr long long CL_NT_STREAM_UNITS = 16ll << 20;      // ... and the smallest streaming-regime launch that takes it again

template <int VEC> struct Vec;
template <> struct Vec<1> { using type = float; };
template <> struct Vec<2> { using type = float __attribute__((ext_vector_type(2))); };
template <> struct Vec<4> { using type = float __attribute__((ext_vector_type(4))); };

template <int VEC>
CL_DEV void vload(float (&dst)[VEC], const float* __restrict__ p) {
    using V = typename Vec<VEC>::type;
    const V v = *reinterpret_cast<const V*>(p);
    if constexpr (VEC == 1) dst[0] = v;
    else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) dst[i] = v[i];
    }
}

// Load from a state / action plane in HBM; NT as in pstore (measured together with the nt stores on cl_step_lean_kernel in round 2,
// 17 x 65 536: 7.32 -> 7.18 us; the copy-floor pattern of scripts/launch_gap.py: 5.65 -> 5.41 us -- and measured AGAIN in round 5, where
// the hint costs the headline instantiation 1 us: which instantiations keep it is lean_step_body's `NTL`).
template <int VEC, bool NT>
CL_DEV void pload(float (&dst)[VEC], const float* __restrict__ p) {
    if constexpr (NT) {
        using V = typename Vec<VEC>::type;
        const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
        if constexpr (VEC == 1) dst[0] = v;
        else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) dst[i] = v[i];
        }
    } else vload<VEC>(dst, p);
}

template <int VEC>
CL_DEV void vstore(float* __restrict__ p, const float (&src)[VEC]) {
    using V = typename Vec<VEC>::type;
    V v;
    if constexpr (VEC == 1) v = src[0];
    else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = src[i];
    }
    *reinterpret_cast<V*>(p) = v;
}

// Store to a state / output plane in HBM (never LDS).  NT = non-temporal (`global_store ... nt`): the lines stream through the
// XCD's L2 instead of staying dirty in it until the end-of-kernel release writes them back.  scripts/launch_gap.py: for the
// headline shape's access pattern (17.8 MB in, 22.3 MB out) the waves are alive 2.9 us either way, but the gap to the next launch's
// first wave is 3.2 us with plain stores and 2.7 us with nt ones (1.1 us for a read-only kernel, 1.8 us period for an empty one);
// cl_step_lean_kernel 17 x 65 536: 7.91 -> 7.20 us.  With a footprint of the order of the 256 MB Infinity Cache the hint costs bandwidth
// instead (17 x 262 144: +10 %), several times past it it wins again (17 x 1 048 576: 125 -> 115 us): the host sets StepArgs::nt by size.
template <int VEC, bool NT>
CL_DEV void pstore(float* __restrict__ p, const float (&src)[VEC]) {
    using V = typename Vec<VEC>::type;
    V v;
    if constexpr (VEC == 1) v = src[0];
    else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = src[i];
    }
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
    else *reinterpret_cast<V*>(p) = v;
}
struct NtOn { static constexpr bool value = true; };
struct NtOff { static constexpr bool value = false; };

// action element (col, env): coalesced when act_stride_env == 1
template <int VEC>
CL_DEV void load_action(float (&dst)[VEC], const StepArgs& a, int col, int env0) {
    if (col < 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) dst[i] = 0.0f;
        return;
    }
    const float* p = a.actions + (long long)col * a.act_stride_col;
    if (a.act_stride_env == 1) vload<VEC>(dst, p + env0);
    else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) dst[i] = p[(long long)(env0 + i) * a.act_stride_env];
    }
}

#include "cl_flex.h"     // flexible loads (EV chargers, EVs, washing machines): device functions + the stand-alone kernel

constexpr int NQ = CL_NQ;

// District sums over buildings: wave partials -> LDS -> fixed-order serial sum over waves (deterministic).
// `stride` is the building stride used by the caller's loop (needed by the MARL second sweep).
// FOLD: instantiations that are launched building-chunked may finish the chunk sums themselves (a.fused_finish); a template parameter
// because the fold's sixteen loads in flight would otherwise set the register budget of every kernel this is inlined into (the general
// lean kernel went from 52 to 102 VGPRs).
// (streaming KPIs, below)  One more sample of a district series costs ONE round trip: the seven accumulators that move every step are
// fetched together (kpi_series_fetch -- where the caller can, before a barrier it has to wait at anyway), then updated and stored
// (kpi_series_apply).  As `k[..] += ..` statements one after the other -- what rounds 1 - 3 had -- every load waits for the store in
// front of it (the compiler cannot tell the planes apart): five dependent round trips at the very end of every workgroup.
struct KpiSeries { float prev, ramp, dsum, dmax, msum, mmax, amax; };
CL_DEV void kpi_series_fetch(KpiSeries& s, const float* __restrict__ k, long long n_env);
CL_DEV void kpi_series_apply(float* __restrict__ k, long long n_env, int t, float v, const KpiSeries& s);
CL_DEV void kpi_series_update(float* __restrict__ k, long long n_env, int t, float v);

// DEFERRED FINISH (cl_tuning.finish = 3, a.fused_finish == 2), the fold half: every workgroup of a building-chunked launch adds up
// `opw` <= 16 district sums (quantity, env) of the PREVIOUS step out of scratch buffer (t + 1) & 1 -- n_chunks x opw partial sums -- in
// cl_finish_kernel's association and writes them to out_env.  What was tried, on the 1024 x 1024 shards (rocprofv3 averages of the step
// kernel, thermal / battery + PV; 11.5 / 7.8 us without any fold):
//  1. one district sum per wave, lane = chunk (a 64-line gather), fetched inside district_reduce behind the plane stores: 13.5 us -- on
//     gfx9 vector loads and stores share one counter and return out of order with respect to each other, so a load waited for after
//     stores were issued waits for their acknowledgements too;
//  2. the same gather fetched at the very top of the kernel: 13.9 / 9.7 us;  3. fetched behind the first building's plane loads and
//     consumed before its stores: 14.1 / 9.8 us -- so not a placement problem: the sixteen waves of a workgroup each gathered 4 bytes
//     out of the SAME 64 lines (16 consecutive envs of 64 chunk rows), 262 144 line requests per launch for 1 MB of data;
//  4. (this one) every line is requested once: thread i of the workgroup fetches partial sum (chunk i / 16, district sum i % 16) -- a
//     wave reads four 64-byte segments -- issued behind the first building's plane loads, parked in LDS before the wave's first store,
//     and added up behind district_reduce's first barrier: wave w takes district sum w, lanes 0..15 form the sixteen wave-partials of
//     cl_finish_kernel (chunks k, k + 16, k + 32, k + 48), then their sum in order: 14.4 / 8.8 us.  Piece by piece on one box
//     (profiles/r04_c4_fold_breakdown.log; 13.0 / 7.8 us with the second launch, whose own 4 - 5 us mostly overlap the next step):
//     double-buffered rows + marker only 13.7 / 8.1 -- step launches now run back to back, each starting into the previous one's
//     draining stores; + load and stash 14.9 / 8.4; + add-up without the load 14.4 / 8.6; everything 15.3 / 9.0.
// (`bx`, `by`: the workgroup's env tile and building chunk -- blockIdx.x / .y, or swapped: see CL_SWAP below)
// (round 5) The exchange tile is [n_chunks][W] with W = 16, 32 or 64 district sums per workgroup row (`opw` rounded up; n_chunks x W <= 1024 =
// one value per thread of the 16-wave workgroup: host): launches with FEWER, LARGER chunks defer too -- the whole BASELINE config 4 on one GPU
// runs 8 chunks of 128 buildings (91 vs 103 us with 32 chunks of 32), where a workgroup row folds 64 sums, four per wave.
CL_DEV int fold_shift(int opw) { return opw <= 16 ? 4 : opw <= 32 ? 5 : 6; }
template <int TILE>
CL_DEV float fold_prefetch(const StepArgs& a, int w, int lane, long long plane, int bx, int by) {
    if (a.fused_finish != 2) return 0.0f;
    const int opw = (NQ * TILE + a.n_chunks - 1) / a.n_chunks;           // district sums of an env tile per workgroup row (<= 64: host)
    const int sh = fold_shift(opw);
    const int i = w * 64 + lane, chunk = i >> sh, j = i & ((1 << sh) - 1), o = by * opw + j;
    const int e = bx * TILE + o % TILE;
    if (chunk < a.n_chunks && j < opw && o < NQ * TILE && e < a.n_env)
        return a.out_bldg[(long long)CLO_RESERVED * plane + ((long long)((a.t + 1) & 1) * a.n_chunks + chunk) * NQ * a.n_env + (long long)(o / TILE) * a.n_env + e];
    return 0.0f;
}
CL_DEV void fold_stash(const StepArgs& a, float* lds_fold, int w, int lane, float v) {
    if (a.fused_finish == 2 && w < 16) lds_fold[w * 64 + lane] = v;        // [n_chunks][W district sums]
}
template <int TILE>
CL_DEV void fold_finish(const StepArgs& a, const float* lds_fold, int w, int lane, int bx, int by) {
    const int opw = (NQ * TILE + a.n_chunks - 1) / a.n_chunks;
    const int sh = fold_shift(opw);
    const int k = lane & 15;
    for (int j = w; j < opw; j += 16) {                                  // wave-uniform: district sums w, w + 16, ... of the row
        const int o = by * opw + j;
        if (o >= NQ * TILE || bx * TILE + o % TILE >= a.n_env) continue;
        float pk = 0.0f;
        if (k < a.n_chunks) {
#pragma unroll
            for (int c = 0; c < 4; ++c) pk += k + 16 * c < a.n_chunks ? lds_fold[((k + 16 * c) << sh) + j] : 0.0f;
        }
        float tot = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pk), q));
        if (lane == 0) a.out_env[(long long)(o / TILE) * a.n_env + bx * TILE + o % TILE] = tot;
    }
}

// KPIS: the thread that writes an env's district net also feeds it to the env's streaming district accumulators (CLD_KPI, lean districts)
// PRESTORED: the caller has kept the wave's partial sums in its LDS row all along (cl_full.h, the C4 shard's kernel: four accumulators fewer in
// registers across the buildings of a wave) -- q_* are not read.
// SWAP (round 5, measured and NOT the default: build with -DCL_SWAP_GRID=true): the launch's grid is (building chunks, env tiles) instead of
// (env tiles, building chunks).  Workgroups go to the eight XCDs round-robin in x-major order: with the env tiles along x every XCD steps one
// tile of EVERY chunk and fetches every building's parameter block and table row into its own L2 (C4 shard: 8 x 327 KB per step); with the
// chunks along x an XCD owns a few chunks for all their env tiles.  The counters confirm the traffic -- C4 shard 64.68 -> 62.46 MB per step =
// 1.039 x algorithmic -- and the clock says no: alternating builds on one box (profiles/r05e_*), thermal shard 13.12 - 13.15 us tile-major
// against 13.66 - 13.76 us chunk-major, battery + PV shard 8.80 - 8.84 against 9.00 - 9.75 us, the whole thermal config (1024 x 8192) 101.0
// against 109.9 us.  An XCD that owns four chunks touches 128 building rows of every plane instead of all 1024: its requests crowd a
// fraction of the rows the memory system interleaves over, and 2 MB of parameter re-reads (L2 misses that the Infinity Cache serves) were
// never on the critical path.
template <int VEC, bool FLEX = false, bool FOLD = false, bool KPIS = false, bool PRESTORED = false, bool SWAP = false>
CL_DEV void district_reduce(const StepArgs& a, float* lds, int w, int lane, int env0, bool live, long long plane, int rkind,
                            const float (&q_net)[VEC], const float (&q_cost)[VEC], const float (&q_em)[VEC],
                            const float (&q_rw)[VEC], int stride, [[maybe_unused]] const float* lds_fold = nullptr) {
    constexpr int TILE = 64 * VEC;
    const int bx = SWAP ? blockIdx.y : blockIdx.x, by = SWAP ? blockIdx.x : blockIdx.y;       // env tile, building chunk
    if constexpr (!PRESTORED) {
        float* mine = lds + (size_t)w * NQ * TILE + lane * VEC;
        vstore<VEC>(mine + 0 * TILE, q_net);
        vstore<VEC>(mine + 1 * TILE, q_cost);
        vstore<VEC>(mine + 2 * TILE, q_em);
        vstore<VEC>(mine + 3 * TILE, q_rw);
    }
    const int tile_env0 = bx * TILE;
    // (KPIS) the control series' accumulators of the env whose district net this thread is about to write: in flight across the barrier
    [[maybe_unused]] KpiSeries pre;
    if constexpr (KPIS) {
        if (threadIdx.x < TILE && tile_env0 + (int)threadIdx.x < a.n_env) kpi_series_fetch(pre, a.kpi_env + tile_env0 + threadIdx.x, a.n_env);
    }
    __syncthreads();
    const bool coupled = rkind == CLR_MARL || (FLEX && rkind == CLR_EV);   // rewards that need the district net
    if (a.n_chunks > 1) {
        // Large districts: this workgroup only saw buildings [y*b_chunk, (y+1)*b_chunk).  Its partial sums go to the scratch rows of
        // out_bldg's reserved plane; cl_finish_kernel (a second launch, 4.8 us of pure latency) adds the chunks in a fixed order.
        // The alternative -- the LAST chunk of an env tile to arrive folds them inside this launch -- is built (FOLD, cl_tuning.finish = 2)
        // and measured SLOWER.  Round 1 did it with an agent-scope release / acquire pair around a counter: those fences write back and
        // invalidate the whole XCD L2 (177 us vs 21 us at 1024 buildings x 1024 envs).  Round 3 removed every fence: only the partial sums
        // and the ticket cross XCDs, so only THEY are accessed at agent scope (relaxed atomic stores / loads = write-through / L2-bypassing
        // `sc1` accesses, no cache maintenance); the order "partials complete -> ticket" is a wait for the thread's own stores (a
        // workgroup-scope release fence is an s_waitcnt) plus the workgroup barrier in front of the one thread that takes the ticket; the
        // fold itself has every thread's loads in flight at once.  Correct and deterministic (scripts/finish_stress.py: 3000 steps, two
        // engines bit-identical) -- and 16.2 us against 14.6 us for the two launches at 1024 x 1024 thermal, 16.0 against 10.5 us for
        // battery + PV (profiles/r03b_c4_fold_ab.log): write-through acknowledgement, ticket and re-read are three dependent device-scope
        // round trips of ~2 us each behind the last workgroup, more than the launch they replace.  The second launch stays the default.
        // DEFERRED FINISH (FOLD, cl_tuning.finish = 3, a.fused_finish == 2; round 4): nothing in step t + 1 reads step t's district sums
        // unless the reward couples the buildings (MARL, EV) -- so the second launch leaves the per-step critical path altogether.  The
        // scratch rows are double-buffered by step parity: this launch writes its partial sums to buffer t & 1 and every wave folds ONE
        // district sum of the PREVIOUS step out of buffer (t + 1) & 1 (plain loads of what the previous launch wrote -- no cross-workgroup
        // traffic inside a launch, no fence, no ticket), in cl_finish_kernel's summation order: out_env trails the step by one launch, and
        // cl_finish_f32 (the same cl_finish_kernel, on buffer t & 1) brings it up to date when somebody wants to read it -- the host's
        // lazy read, or the end of a captured rollout.  A marker word per buffer (step + 1; with the chunk count, in the last 16 bytes of the
        // reserved plane) tells cl_finish_f32 whether the buffer really holds step t's partial sums: calls that did not defer leave it
        // alone, and cl_finish_f32 is then a no-op -- the library keeps no host-side state about pending folds.
        float* scratch = a.out_bldg + (long long)CLO_RESERVED * plane;
        const bool deferred = FOLD && a.fused_finish == 2;
        if (deferred) scratch += (long long)(a.t & 1) * a.n_chunks * NQ * a.n_env;
        for (int i = threadIdx.x; i < NQ * TILE; i += blockDim.x) {
            const int q = i / TILE, e = i - q * TILE;
            float s = 0.0f;
            for (int k = 0; k < a.nw; ++k) s += lds[(size_t)k * NQ * TILE + i];
            if (tile_env0 + e < a.n_env) {
                float* dst = scratch + ((long long)by * NQ + q) * a.n_env + tile_env0 + e;
                if (FOLD && a.fused_finish == 1) __hip_atomic_store(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *dst = s;
            }
        }
        if constexpr (!FOLD) return;                             // cl_finish_kernel folds them (second launch)
        else {
        if (!a.fused_finish) return;                             // cl_tuning.finish = 1: likewise
        // (a launch that does NOT defer clears this step's marker -- here for the in-launch fold, in cl_finish_kernel for the second launch:
        //  a buffer left behind by an earlier deferred step with the same parity must not look current to a later cl_finish_f32)
        if (a.fused_finish == 1 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
            reinterpret_cast<unsigned*>(a.out_bldg + (long long)(CLO_RESERVED + 1) * plane - 4)[a.t & 1] = 0u;
        if (a.fused_finish == 2) {
            // the previous step's district sums, behind this step's partial-sum stores (the exchange tile was complete at the barrier above)
            fold_finish<TILE>(a, lds_fold, w, lane, bx, by);
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
                unsigned* marker = reinterpret_cast<unsigned*>(a.out_bldg + (long long)(CLO_RESERVED + 1) * plane - 4);   // last 16 bytes of the plane
                marker[a.t & 1] = (unsigned)a.t + 1u;
                marker[2] = (unsigned)a.n_chunks;
            }
            return;
        }
        // (the tickets sit at a FIXED place -- the (n_env + 63) / 64 words in front of the marker words at the plane's tail -- whatever the chunk
        //  count: behind the partial-sum rows, where they were until round 5, a chunked cl_rollout_f32 with another chunk geometry wrote its
        //  return rows over them, and a later in-launch fold found non-zero tickets and never folded: round-5 advisor finding)
        unsigned* ticket = reinterpret_cast<unsigned*>(a.out_bldg + (long long)(CLO_RESERVED + 1) * plane) - 4 - (a.n_env + 63) / 64 + bx;
        // this thread's partial sums have left the CU: outside tgsplit mode a workgroup-scope release fence does NOT wait for outstanding
        // vector stores (it only orders LDS), so the wait is spelled out -- s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0) -- in front of it
        // (round-3 advisor finding: the ticket could otherwise become visible before the write-through partial sums are acknowledged)
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();                                         // ... and so have everybody else's (and nobody reads the wave rows of `lds` any more)
        unsigned* flag = reinterpret_cast<unsigned*>(lds);
        if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != (unsigned)a.n_chunks - 1u) return;
        __syncthreads();                                         // (everybody has read the flag: `lds` is reused below)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // The fold is one dependent chain per thread -- ticket, loads, adds -- so its cost is memory round trips, not bytes (32 chunks x
        // 4 quantities x 128 envs = 64 KB).  Every thread of the workgroup takes part: output o = (quantity, env) is shared by
        // `rep` = blockDim / (NQ * TILE) threads, each adding a contiguous range of chunks with all its loads in flight at once (sixteen
        // at a time); the ranges are then added in range order through LDS: the same fixed association whoever arrives last.
        const bool marl = rkind == CLR_MARL;
        const int n_out = NQ * TILE;
        const int rep = max(1, (int)blockDim.x / n_out);
        const int cpp = (a.n_chunks + rep - 1) / rep;                       // chunks per range
        float* part = lds;                                                  // [rep][n_out] (and [rep][n_out] district-net parts for MARL behind it)
        for (int i = threadIdx.x; i < rep * n_out; i += blockDim.x) {
            const int o = i % n_out, p = i / n_out;
            const int q = o / TILE, e = o - q * TILE;
            const bool scale = marl && q == CLQ_REWARD;          // the partials carried sign(-net) * 0.01 * net^2: times max(0, district net)
            float s = 0.0f, sn = 0.0f;
            if (tile_env0 + e < a.n_env) {
                const float* src = scratch + (long long)q * a.n_env + tile_env0 + e;
                const float* srcn = scratch + (long long)CLQ_NET * a.n_env + tile_env0 + e;
                const int c_hi = min(a.n_chunks, (p + 1) * cpp);
                for (int c0 = p * cpp; c0 < c_hi; c0 += 16) {
                    float v[16], vn[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const bool in = c0 + j < c_hi;
                        const long long off = (long long)(in ? c0 + j : c0) * NQ * a.n_env;
                        v[j] = __hip_atomic_load(src + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        vn[j] = scale ? __hip_atomic_load(srcn + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
                        if (!in) { v[j] = 0.0f; vn[j] = 0.0f; }
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) { s += v[j]; sn += vn[j]; }
                }
            }
            part[(size_t)p * n_out + o] = s;
            if (marl) part[(size_t)(rep + p) * n_out + o] = sn;
        }
        __syncthreads();
        for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
            const int q = o / TILE, e = o - q * TILE;
            if (tile_env0 + e >= a.n_env) continue;
            float s = 0.0f, sn = 0.0f;
            for (int p = 0; p < rep; ++p) {
                s += part[(size_t)p * n_out + o];
                if (marl) sn += part[(size_t)(rep + p) * n_out + o];
            }
            if (marl && q == CLQ_REWARD) s *= fmaxf(0.0f, sn);
            a.out_env[(long long)q * a.n_env + tile_env0 + e] = s;
        }
        if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        return;
        }
    }
    for (int i = threadIdx.x; i < NQ * TILE; i += blockDim.x) {
        const int q = i / TILE, e = i - q * TILE;
        float s = 0.0f;
        for (int k = 0; k < a.nw; ++k) s += lds[(size_t)k * NQ * TILE + i];
        if (coupled && q == CLQ_REWARD) {                      // finished below
            if (FLEX && rkind == CLR_EV) lds[i] = s;           // CLR_EV accumulated sign(-net) * 0.01 * net^2: the district MARL sum / max(0, district net)
            continue;
        }
        if (tile_env0 + e < a.n_env) {
            a.out_env[(long long)q * a.n_env + tile_env0 + e] = s;
            if constexpr (KPIS) {
                if (q == CLQ_NET) {                                                                   // control condition (citylearn.py:1136-1323)
                    if (i == (int)threadIdx.x) kpi_series_apply(a.kpi_env + tile_env0 + e, a.n_env, a.t, s, pre);
                    else kpi_series_update(a.kpi_env + tile_env0 + e, a.n_env, a.t, s);               // (workgroups narrower than their env tile)
                }
            }
        }
        if (coupled && q == CLQ_NET) lds[i] = s;               // wave-0 slot now holds the district net
    }
    if (coupled) {
        // MARL couples every building to the district net (reward_function.py:132-143): second sweep over the
        // nets this same thread wrote a moment ago (L1/L2 hits), then a second LDS reduction for the reward sum.
        __syncthreads();
        float dnet[VEC], dmarl[VEC];
        vload<VEC>(dnet, lds + CLQ_NET * TILE + lane * VEC);
        const bool central_ev = FLEX && rkind == CLR_EV && (a.flags & CLD_CENTRAL_AGENT);
        if (central_ev) {
            vload<VEC>(dmarl, lds + CLQ_REWARD * TILE + lane * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) dmarl[i] *= fmaxf(0.0f, dnet[i]);
        }
        __syncthreads();
        float r_sum[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) r_sum[i] = 0.0f;
        if (live) {
            for (int b = w; b < a.n_bldg; b += stride) {
                const long long off = (long long)b * (FLEX ? a.n_env : a.ld) + env0;
                float n[VEC], rw[VEC];
                vload<VEC>(n, a.out_bldg + CLO_NET * plane + off);
#pragma unroll
                for (int i = 0; i < VEC; ++i) rw[i] = central_ev ? dmarl[i] : cl::marl_reward(n[i], dnet[i]);
                if (FLEX && rkind == CLR_EV) {
                    // Electric_Vehicles_Reward_Function: MARL only scales the charger terms cl_flex_kernel prepared
                    const uint32_t* __restrict__ bp = a.params + (long long)b * CL_NP;
                    const int fbi = (bp[CLP_L_FLAGS] & CLF_FLEX) ? (int)bp[CLP_FLEX_INDEX] : -1;
                    float k0[VEC], kn[VEC], kp[VEC];
                    if (fbi >= 0) {
                        const long long fp = (long long)a.n_flex_bldg * a.n_env, fo = (long long)fbi * a.n_env + env0;
                        vload<VEC>(k0, a.flex_out + CLX_RW_K0 * fp + fo);
                        vload<VEC>(kn, a.flex_out + CLX_RW_KNEG * fp + fo);
                        vload<VEC>(kp, a.flex_out + CLX_RW_KPOS * fp + fo);
                    }
#pragma unroll
                    for (int i = 0; i < VEC; ++i) rw[i] = fbi >= 0 ? cl::ev_reward(true, rw[i], n[i], k0[i], kn[i], kp[i]) : 0.0f;
                    if (fbi >= 0 && a.ev_penalty_coef > 0.0f) {            // reward_function.py:431-434
                        float viol[VEC];
                        vload<VEC>(viol, a.flex_out + CLX_VIOLATION * ((long long)a.n_flex_bldg * a.n_env) + (long long)fbi * a.n_env + env0);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) rw[i] -= viol[i] > 0.0f ? viol[i] * a.ev_penalty_coef : 0.0f;
                    }
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) r_sum[i] += rw[i];
                pstore<VEC, false>(a.out_bldg + CLO_REWARD * plane + off, rw);
            }
        }
        vstore<VEC>(lds + (size_t)w * TILE + lane * VEC, r_sum);
        __syncthreads();
        for (int e = threadIdx.x; e < TILE; e += blockDim.x) {
            float s = 0.0f;
            for (int k = 0; k < a.nw; ++k) s += lds[(size_t)k * TILE + e];
            if (tile_env0 + e < a.n_env) a.out_env[(long long)CLQ_REWARD * a.n_env + tile_env0 + e] = s;
        }
    }
}

// FLEX: the district has EV chargers / washing machines (cl_flex.h ran just before); a separate instantiation so that
// districts without them keep their register budget.
// PREC: the battery map -- 0 fp32, 1 CLD_F64_MAPS (two more state planes), 2 CLD_F64_CHAIN (cl_unit.h)
// CHECK (CLD_CHECK): the reference's runtime assertions as CLV_* bits, one word per unit, into the reserved plane (never a chunked launch)
template <int VEC, bool FULL, bool DETAIL, bool FLEX = false, int PREC = 0, bool FOLD = false, bool CHECK = false>
__global__ void __launch_bounds__(1024) cl_step_kernel(const StepArgs a) {
    constexpr bool F64 = PREC == 1;
    constexpr bool SWAP = FOLD && CL_SWAP_GRID;            // (the FOLD instantiations are always launched building-chunked: grid = (chunks, env tiles))
    const int bx = SWAP ? blockIdx.y : blockIdx.x, by = SWAP ? blockIdx.x : blockIdx.y;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [nw][NQ][64*VEC]
    constexpr int TILE = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int env0 = bx * TILE + lane * VEC;
    const bool live = env0 < a.n_env;                     // n_env % VEC == 0 is enforced on the host
    // (a row pitch exists for battery + PV districts only: the thermal / flexible-load instantiations never read it -- their scalar
    //  register file is full, one more live word costs them a scratch reservation)
    const int ld = (FULL || FLEX) ? a.n_env : a.ld;
    const long long plane = (long long)a.n_bldg * ld;
    const int rkind = (a.flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT;
    const bool quirk = a.flags & CLD_REF_T0_QUIRK;
    constexpr bool detail = DETAIL;

    float q_net[VEC], q_cost[VEC], q_em[VEC], q_rw[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) q_net[i] = q_cost[i] = q_em[i] = q_rw[i] = 0.0f;

    const int b_lo = by * a.b_chunk;
    const int b_hi = min(a.n_bldg, b_lo + a.b_chunk);
    const bool marl_partial = rkind == CLR_MARL && a.n_chunks > 1;
    // table row of this env tile: TILE divides CL_ROW0_BLOCK, so the offset is workgroup-uniform (scalar load)
    const int ts_row = a.t + (a.env_row0 ? a.env_row0[(bx * TILE) / CL_ROW0_BLOCK] : 0);
    // (FOLD, deferred finish) this wave's share of the previous step's chunk sums: the first load of the kernel
    [[maybe_unused]] float fold_prev = 0.0f;
    [[maybe_unused]] bool fold_issued = false, folded = false;
    [[maybe_unused]] float* lds_fold = lds + (size_t)a.nw * NQ * TILE;        // (deferred finish) [64][16] behind the reduction rows
    for (int b = b_lo + w; b < b_hi; b += a.nw) {
        cl::Bp B;
        cl::load_bp<FULL>(B, a.params + (long long)b * CL_NP);
        cl::Row R;
        cl::load_row<FULL>(R, a.ts + ((long long)ts_row * a.n_bldg + b) * CL_NF, B.flags,
                           (FULL && DETAIL) ? a.ts + ((long long)(ts_row - a.t + a.n_steps - 1) * a.n_bldg + b) * CL_NF : nullptr);
        if (live) {
            const long long off = (long long)b * ld + env0;
            float s_soc[VEC], s_eff[VEC], s_deg[VEC], s_cs[VEC], s_hs[VEC], s_ds[VEC];
            [[maybe_unused]] float s_efl[VEC], s_dgl[VEC];                 // CLD_F64_MAPS: low words of efficiency / degraded capacity
            float a_cs[VEC], a_hs[VEC], a_ds[VEC], a_es[VEC], a_cd[VEC], a_hd[VEC];
            const bool batt = B.flags & CLF_BATTERY;
            if (batt) {
                vload<VEC>(s_soc, a.state + CLS_B_SOC * plane + off);
                vload<VEC>(s_eff, a.state + CLS_B_EFF * plane + off);
                vload<VEC>(s_deg, a.state + CLS_B_DEGCAP * plane + off);
                if constexpr (F64) {
                    vload<VEC>(s_efl, a.state + CLS_B_EFF_LO * plane + off);
                    vload<VEC>(s_dgl, a.state + CLS_B_DEGCAP_LO * plane + off);
                }
            }
            load_action<VEC>(a_es, a, B.a_es, env0);
            if constexpr (FULL) {
                if (B.flags & CLF_COOL_STO) vload<VEC>(s_cs, a.state + CLS_CS_SOC * plane + off);
                if (B.flags & CLF_HEAT_STO) vload<VEC>(s_hs, a.state + CLS_HS_SOC * plane + off);
                if (B.flags & CLF_DHW_STO) vload<VEC>(s_ds, a.state + CLS_DS_SOC * plane + off);
                load_action<VEC>(a_cs, a, B.a_cs, env0);
                load_action<VEC>(a_hs, a, B.a_hs, env0);
                load_action<VEC>(a_ds, a, B.a_ds, env0);
                if (B.a_coh >= 0) {
                    float c[VEC];
                    load_action<VEC>(c, a, B.a_coh, env0);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) { a_cd[i] = fabsf(fminf(c[i], 0.0f)); a_hd[i] = fabsf(fmaxf(c[i], 0.0f)); }
                } else {
                    load_action<VEC>(a_cd, a, B.a_cd, env0);
                    load_action<VEC>(a_hd, a, B.a_hd, env0);
                }
            }
            if constexpr (FOLD) {
                // (deferred finish) this wave's share of the previous step's chunk sums: issued BEHIND the first building's plane loads --
                // it crosses XCDs (another workgroup's L2 wrote it) and returns later than they do, and loads return in order
                if (!fold_issued) { fold_prev = fold_prefetch<TILE>(a, w, lane, plane, bx, by); fold_issued = true; }
            }
            float o_net[VEC], o_rw[VEC], o_eb[VEC], o_cd[VEC], o_hd[VEC], o_dd[VEC], o_cc[VEC], o_ch[VEC], o_cw[VEC], o_cn[VEC], o_bn[VEC], o_ex[VEC], o_sv[VEC], o_ws[VEC], o_sc[VEC], o_sh[VEC], o_sd[VEC];
            [[maybe_unused]] float o_viol[VEC];
            // chargers / washing machines of this building, advanced by cl_flex_kernel just before this launch
            const int fbi = (FLEX && (B.flags & CLF_FLEX)) ? (int)B.p[CLP_FLEX_INDEX] : -1;
            float x_load[VEC], x_chg[VEC];
            if (FLEX && fbi >= 0) {
                const long long fp = (long long)a.n_flex_bldg * a.n_env, fo = (long long)fbi * a.n_env + env0;
                vload<VEC>(x_load, a.flex_out + CLX_LOAD * fp + fo);
                if constexpr (DETAIL) vload<VEC>(x_chg, a.flex_out + CLX_CHARGERS * fp + fo);   // only the baseline needs it
                else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) x_chg[i] = 0.0f;
                }
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                cl::State S;
                S.soc = batt ? s_soc[i] : 0.0f; S.eff = batt ? s_eff[i] : 1.0f; S.degcap = batt ? s_deg[i] : 0.0f;
                S.cs = S.hs = S.ds = 0.0f;
                S.eff_lo = S.deg_lo = 0.0f;
                if constexpr (F64) { S.eff_lo = batt ? s_efl[i] : 0.0f; S.deg_lo = batt ? s_dgl[i] : 0.0f; }
                cl::Act act = {0.0f, 0.0f, 0.0f, a_es[i], 0.0f, 0.0f};
                if constexpr (FULL) {
                    S.cs = (B.flags & CLF_COOL_STO) ? s_cs[i] : 0.0f;
                    S.hs = (B.flags & CLF_HEAT_STO) ? s_hs[i] : 0.0f;
                    S.ds = (B.flags & CLF_DHW_STO) ? s_ds[i] : 0.0f;
                    act = {a_cs[i], a_hs[i], a_ds[i], a_es[i], a_cd[i], a_hd[i]};
                }
                cl::Out O;
                cl::unit_step<FULL, PREC, CHECK>(B, R, a.t, quirk, act, S, O);
                if constexpr (CHECK) o_viol[i] = __uint_as_float(O.viol);
                if (FLEX && fbi >= 0) cl::apply_flex(R.outage, R.price, R.carbon, x_load[i], x_chg[i], O);
                const float rw = cl::unit_reward<FULL>(rkind, B, S, O.net);
                s_soc[i] = S.soc; s_eff[i] = S.eff; s_deg[i] = S.degcap; s_cs[i] = S.cs; s_hs[i] = S.hs; s_ds[i] = S.ds;
                if constexpr (F64) { s_efl[i] = S.eff_lo; s_dgl[i] = S.deg_lo; }
                o_net[i] = O.net; o_rw[i] = rw; o_eb[i] = O.eb; o_cd[i] = O.cool_dem; o_hd[i] = O.heat_dem; o_dd[i] = O.dhw_dem;
                o_cc[i] = O.c_cool; o_ch[i] = O.c_heat; o_cw[i] = O.c_dhw; o_cn[i] = O.c_ns;
                o_bn[i] = O.base_net; o_ex[i] = O.expected; o_sv[i] = O.served; o_ws[i] = O.net_ws;
                o_sc[i] = O.se_cool; o_sh[i] = O.se_heat; o_sd[i] = O.se_dhw;
                // multi-chunk MARL: accumulate sign(-net) * 0.01 * net^2; cl_finish_kernel scales by max(0, district net)
                q_net[i] += O.net; q_cost[i] += O.cost; q_em[i] += O.emission;
                q_rw[i] += (marl_partial || (FLEX && rkind == CLR_EV)) ? cl::marl_reward(O.net, 1.0f) : rw;
            }
            if constexpr (FOLD) {
                // ... and parked in LDS after the LAST building's arithmetic, before its stores: the load crosses XCDs and takes longer than one
                // building's arithmetic (stash before the first of two buildings' stores: + 1.1 us on the thermal shard); the stores of earlier
                // buildings have long been acknowledged by then (a load waited for right behind stores waits for their acknowledgements too)
                if (!folded && b + a.nw >= b_hi) { fold_stash(a, lds_fold, w, lane, fold_prev); folded = true; }
            }
            auto put = [&](auto nt_tag) {
                constexpr bool NT = decltype(nt_tag)::value;
                if (batt) {
                    pstore<VEC, NT>(a.state + CLS_B_SOC * plane + off, s_soc);
                    pstore<VEC, NT>(a.state + CLS_B_EFF * plane + off, s_eff);
                    pstore<VEC, NT>(a.state + CLS_B_DEGCAP * plane + off, s_deg);
                    if constexpr (F64) {
                        pstore<VEC, NT>(a.state + CLS_B_EFF_LO * plane + off, s_efl);
                        pstore<VEC, NT>(a.state + CLS_B_DEGCAP_LO * plane + off, s_dgl);
                    }
                }
                if constexpr (FULL) {
                    if (B.flags & CLF_COOL_STO) pstore<VEC, NT>(a.state + CLS_CS_SOC * plane + off, s_cs);
                    if (B.flags & CLF_HEAT_STO) pstore<VEC, NT>(a.state + CLS_HS_SOC * plane + off, s_hs);
                    if (B.flags & CLF_DHW_STO) pstore<VEC, NT>(a.state + CLS_DS_SOC * plane + off, s_ds);
                }
                pstore<VEC, NT>(a.out_bldg + CLO_NET * plane + off, o_net);
                if (rkind != CLR_MARL && !(FLEX && rkind == CLR_EV)) pstore<VEC, NT>(a.out_bldg + CLO_REWARD * plane + off, o_rw);
                if constexpr (CHECK) pstore<VEC, NT>(a.out_bldg + CLO_RESERVED * plane + off, o_viol);
                if constexpr (FULL && DETAIL) {
                    // what another kernel of the path reads: the KPI pass (baseline, expected, served) and the LSTM stage (delivered demands)
                    pstore<VEC, NT>(a.out_bldg + CLO_COOL_DEM * plane + off, o_cd);
                    pstore<VEC, NT>(a.out_bldg + CLO_HEAT_DEM * plane + off, o_hd);
                    pstore<VEC, NT>(a.out_bldg + CLO_BASE_NET * plane + off, o_bn);
                    pstore<VEC, NT>(a.out_bldg + CLO_EXPECTED * plane + off, o_ex);
                    pstore<VEC, NT>(a.out_bldg + CLO_SERVED * plane + off, o_sv);
                    if (!(a.flags & CLD_DETAIL_MIN)) {                       // the series of evaluate() / the observations / the parity tests
                        pstore<VEC, NT>(a.out_bldg + CLO_B_EB * plane + off, o_eb);
                        pstore<VEC, NT>(a.out_bldg + CLO_DHW_DEM * plane + off, o_dd);
                        pstore<VEC, NT>(a.out_bldg + CLO_C_COOL * plane + off, o_cc);
                        pstore<VEC, NT>(a.out_bldg + CLO_C_HEAT * plane + off, o_ch);
                        pstore<VEC, NT>(a.out_bldg + CLO_C_DHW * plane + off, o_cw);
                        pstore<VEC, NT>(a.out_bldg + CLO_C_NSL * plane + off, o_cn);
                        pstore<VEC, NT>(a.out_bldg + CLO_NET_WS * plane + off, o_ws);
                        pstore<VEC, NT>(a.out_bldg + CLO_SE_COOL * plane + off, o_sc);
                        pstore<VEC, NT>(a.out_bldg + CLO_SE_HEAT * plane + off, o_sh);
                        pstore<VEC, NT>(a.out_bldg + CLO_SE_DHW * plane + off, o_sd);
                    }
                }
            };
            if (a.nt) put(NtOn{}); else put(NtOff{});
        }
    }

    if constexpr (FOLD) {
        if (!folded) {                                              // (waves without a building or beyond the batch)
            if (!fold_issued) fold_prev = fold_prefetch<TILE>(a, w, lane, plane, bx, by);
            fold_stash(a, lds_fold, w, lane, fold_prev);
        }
    }
    district_reduce<VEC, FLEX, FOLD, false, false, SWAP>(a, lds, w, lane, env0, live, plane, rkind, q_net, q_cost, q_em, q_rw, a.nw, lds_fold);
}

// Building-chunked battery + PV districts (BASELINE config 4 with the 2022 device set), latency-ordered (round 5).  cl_step_kernel walks a
// dependent chain per building -- parameter block (scalar loads) -> battery flag / action column -> plane loads -> arithmetic -> stores --
// and a wave of a chunked launch walks it b_chunk / 16 times.  Here a wave issues the plane loads of its FIRST building before it touches a
// parameter and those of its NEXT building before it computes the current one: the addresses depend on the building index only (and on
// CLD_ES_COL_IS_BLDG for the action column; other column maps load the action once the block is there).  The parameter block and the table
// row still arrive by scalar loads (staged in LDS they would sit in 36 vector registers per lane: 44 bytes of scratch at four envs per lane).
// Non-battery buildings' planes are fetched and ignored.  Same arithmetic, same summation order as cl_step_kernel<VEC, false, false>:
// bit-identical planes and sums.
// PREC = 2 (round 6): the battery's soc chain in float64 (CLD_F64_CHAIN, the engine's default precision model) -- the chunked districts' default path
// had been the general kernel plus the second launch.
template <int VEC, bool NT, bool FOLD, int PREC = 0>
__global__ void __launch_bounds__(1024) cl_step_lean_chunk_kernel(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [nw][NQ][64*VEC] | (FOLD) exchange tile
    constexpr int TILE = 64 * VEC;
    const int bx = blockIdx.x, by = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int env0 = bx * TILE + lane * VEC;
    const bool live = env0 < a.n_env;
    const int ld = a.n_env;                                       // (no row pitch for districts of more than 32 buildings: host)
    const long long plane = (long long)a.n_bldg * ld;
    const int rkind = (a.flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT;
    const bool quirk = a.flags & CLD_REF_T0_QUIRK;
    const bool act_by_bldg = (a.flags & CLD_ES_COL_IS_BLDG) && a.act_stride_env == 1;
    const int b_lo = by * a.b_chunk;
    const int b_hi = min(a.n_bldg, b_lo + a.b_chunk);
    const bool marl_partial = rkind == CLR_MARL;                  // (always chunked: cl_finish_kernel scales by the district net)
    const int ts_row = a.t + (a.env_row0 ? a.env_row0[(bx * TILE) / CL_ROW0_BLOCK] : 0);
    [[maybe_unused]] float* lds_fold = lds + (size_t)a.nw * NQ * TILE;
    struct In { float soc[VEC], eff[VEC], deg[VEC], act[VEC]; };
    auto fetch = [&](In& x, int b) {
        const long long off = (long long)b * ld + env0;
        vload<VEC>(x.soc, a.state + CLS_B_SOC * plane + off);
        vload<VEC>(x.eff, a.state + CLS_B_EFF * plane + off);
        vload<VEC>(x.deg, a.state + CLS_B_DEGCAP * plane + off);
        if (act_by_bldg) vload<VEC>(x.act, a.actions + (long long)b * a.act_stride_col + env0);
    };
    In cur;
#pragma unroll
    for (int i = 0; i < VEC; ++i) cur.soc[i] = cur.eff[i] = cur.deg[i] = cur.act[i] = 0.0f;
    int b = b_lo + w;
    if (live && b < b_hi) fetch(cur, b);
    [[maybe_unused]] float fold_prev = 0.0f;
    [[maybe_unused]] bool folded = false;
    if constexpr (FOLD) fold_prev = fold_prefetch<TILE>(a, w, lane, plane, bx, by);      // behind the first building's plane loads (fold_prefetch's note)
    float q_net[VEC], q_cost[VEC], q_em[VEC], q_rw[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) q_net[i] = q_cost[i] = q_em[i] = q_rw[i] = 0.0f;
    for (; b < b_hi; b += a.nw) {
        cl::Bp B;
        cl::load_bp<false>(B, a.params + (long long)b * CL_NP);
        cl::Row R;
        cl::load_row<false>(R, a.ts + ((long long)ts_row * a.n_bldg + b) * CL_NF, B.flags);
        if (live) {
            const long long off = (long long)b * ld + env0;
            const bool batt = B.flags & CLF_BATTERY;
            float a_es[VEC];
            if (act_by_bldg) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) a_es[i] = B.a_es >= 0 ? cur.act[i] : 0.0f;
            } else load_action<VEC>(a_es, a, B.a_es, env0);
            float o_soc[VEC], o_eff[VEC], o_deg[VEC], o_net[VEC], o_rw[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                cl::State S;
                S.soc = batt ? cur.soc[i] : 0.0f; S.eff = batt ? cur.eff[i] : 1.0f; S.degcap = batt ? cur.deg[i] : 0.0f;
                S.cs = S.hs = S.ds = 0.0f; S.eff_lo = S.deg_lo = 0.0f;
                const cl::Act act = {0.0f, 0.0f, 0.0f, a_es[i], 0.0f, 0.0f};
                cl::Out O;
                cl::unit_step<false, PREC>(B, R, a.t, quirk, act, S, O);
                const float rw = cl::unit_reward<false>(rkind, B, S, O.net);
                o_soc[i] = S.soc; o_eff[i] = S.eff; o_deg[i] = S.degcap; o_net[i] = O.net; o_rw[i] = rw;
                q_net[i] += O.net; q_cost[i] += O.cost; q_em[i] += O.emission;
                q_rw[i] += marl_partial ? cl::marl_reward(O.net, 1.0f) : rw;
            }
            if constexpr (FOLD) {
                if (!folded && b + a.nw >= b_hi) { fold_stash(a, lds_fold, w, lane, fold_prev); folded = true; }     // before the LAST building's stores
            }
            // the NEXT building's plane loads, into the registers the arithmetic has just released and IN FRONT of this building's stores: the
            // wait for them at the top of the next iteration then allows the stores to be outstanding (loads and stores share one counter)
            if (b + a.nw < b_hi) fetch(cur, b + a.nw);
            if (batt) {
                pstore<VEC, NT>(a.state + CLS_B_SOC * plane + off, o_soc);
                pstore<VEC, NT>(a.state + CLS_B_EFF * plane + off, o_eff);
                pstore<VEC, NT>(a.state + CLS_B_DEGCAP * plane + off, o_deg);
            }
            pstore<VEC, NT>(a.out_bldg + CLO_NET * plane + off, o_net);
            if (rkind != CLR_MARL) pstore<VEC, NT>(a.out_bldg + CLO_REWARD * plane + off, o_rw);
        }
    }
    if constexpr (FOLD) {
        if (!folded) fold_stash(a, lds_fold, w, lane, fold_prev);      // (waves without a building or beyond the batch)
    }
    district_reduce<VEC, false, FOLD>(a, lds, w, lane, env0, live, plane, rkind, q_net, q_cost, q_em, q_rw, a.nw, lds_fold);
}

// Lean districts (battery + PV + load), at most two buildings per wave, one chunk: the headline shape.  Same arithmetic
// as cl_step_kernel<VEC, false, false>; the memory operations are re-ordered for latency.  A kernel with this launch
// shape and these byte counts but no energy model runs in 5.4 us at 17 x 65 536 (scripts/copy_floor.py) against 8.1 us
// for the generic kernel, whose every wave walks a dependent chain  parameter block (scalar) -> action column ->
// state / action loads -> compute -> stores  once per building.  Here a wave issues the state loads of BOTH its
// buildings (and the action loads, when the column of building b is b: CLD_ES_COL_IS_BLDG) before it touches a parameter.
// (Tried: VEC = 2 compiled for 8 waves per SIMD -- two workgroups per CU, half the per-wave chain: 64 VGPRs with 23 spilled,
// 9.3 us vs 7.9 us for VEC = 4 at 17 x 65 536.  Tried: with 17 = 16 + 1 buildings, sharing the last building's env tile out
// in 64-env strips to four waves at one env per lane instead of giving one wave two buildings -- 5 instead of 8 units on the
// longest wave, 62 instead of 94 VGPRs: 7.89 vs 7.94 us, i.e. the doubled wave is not what bounds the launch.  Tried:
// non-temporal stores for the net / reward planes: 7.87 - 7.91 vs 7.86 - 7.88 us, no effect.  Tried (round 2): plane accesses
// through `float` lvalues instead of float4 ones, so that the parameter block of wave 0's SECOND building -- read after the first
// one's stores, and therefore a uniform *vector* load behind them -- stays a scalar load: 73 instead of 88 VGPRs, and 8.14 vs
// 7.99 us, three alternations on one box: slower.  The same trick is what made the thermal kernel's two-env pack viable
// (cl_full.h); here the vector loads of the 17th building's parameters are issued early enough and the scalar ones are not.
//  Tried again at the end of round 2, with the nt stores and the hoisted unit in place: the 17th building shared out to four waves on
//  four SIMDs as 64-env strips at one env per lane (SIMD loads 4.25 each instead of 5, 4, 4, 4; the strips' district partials in
//  their own LDS row, processed before the wave's main building so that its parameter block stays a scalar load): 7.36 - 7.49 vs
//  7.37 - 7.63 us, alternating on one box -- nothing.  scripts/lean_balance_probe.py: 16 buildings x 65 536 take 6.58 us, 17 take
//  7.39 us, 20 (four waves with two buildings: 5, 5, 5, 5) take 8.22 us = 6.98 us per 17: the cost of the 17th building is that
//  SOME wave walks a second dependent chain behind its first, not the imbalance between the SIMDs.)
// The compact observation of the NEXT row written by the step launch itself (cl_step_observe_f32): every column is an affine map
// of a plane value the wave that stepped the building still holds in registers.
struct ObsFusedArgs {
    float* __restrict__ obs;            // [n_env][pitch]
    const float* __restrict__ table;    // obs_table [n_rows][n_cols]: the offsets of the affine maps
    int n_cols, pitch, row;
    short start[CL_OBS_FUSED_BLDG + 1]; // deps[start[b] .. start[b + 1]) belong to building b
    cl_obs_dep deps[CLOB_MAX_DEPS];     // grouped by building
    int lean_ok;                        // (host) the battery + PV launch can fill it: <= CL_OBS_FUSED_PER_BLDG columns per building, battery / net / reward planes only
};

template <int VEC, bool FLEX, bool NT, bool OBS, bool KPI = false, int PREC = 0>
CL_DEV void lean_step_body(const StepArgs& a, float* lds, const ObsFusedArgs* of) {     // lds: [nw][NQ][64*VEC] (, then [64*VEC][pitch])
    constexpr bool F64 = PREC == 1;
    constexpr int TILE = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int env0 = blockIdx.x * TILE + lane * VEC;
    const bool live = env0 < a.n_env;
    const long long plane = (long long)a.n_bldg * a.ld;
    const int rkind = (a.flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT;
    const bool quirk = a.flags & CLD_REF_T0_QUIRK;
    const bool act_by_bldg = (a.flags & CLD_ES_COL_IS_BLDG) && a.act_stride_env == 1;
    const int ts_row = a.t + (a.env_row0 ? a.env_row0[(blockIdx.x * TILE) / CL_ROW0_BLOCK] : 0);
    const int bb[2] = {w, w + a.nw};
    const bool own[2] = {bb[0] < a.n_bldg, bb[1] < a.n_bldg};       // wave-uniform

    float q_net[VEC], q_cost[VEC], q_em[VEC], q_rw[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) q_net[i] = q_cost[i] = q_em[i] = q_rw[i] = 0.0f;
    float s_soc[2][VEC], s_eff[2][VEC], s_deg[2][VEC], a_es[2][VEC];
    [[maybe_unused]] float s_efl[2][VEC], s_dgl[2][VEC];             // CLD_F64_MAPS: low words of efficiency / degraded capacity
    // Non-temporal hint on the plane LOADS (the stores: StepArgs::nt): by instantiation, from alternating A/B builds on one box at 17 buildings
    // (-DCL_LEAN_NT_LOADS=false against the default; profiles/r05_nt_loads/): four envs per lane 65 536 envs 7.52 - 7.56 us with the hint,
    // 6.49 us WITHOUT (49 152: 6.82 / 6.40; 81 920: 11.0 / 10.1) -- the round-2 measurement that introduced the hint (7.32 -> 7.18 us) predates
    // the eight-tensor action ring and the full-year tables; two envs per lane 24 576 / 32 768 envs 4.49 - 4.54 / 4.78 - 4.91 us with, 4.74 - 4.76 /
    // 4.99 - 5.02 without; one env per lane 8 192 / 16 384 envs 3.45 / 4.00 with, 3.34 / 3.78 without; the KPI epilogue's instantiation (four
    // envs per lane) 12.54 us with, 14.78 without.  (With plain STORES the loads' hint is immaterial: 7.29 / 7.31 us.)
    // The flexible-load instantiation (EV districts, behind cl_flex_kernel) keeps it too: 16.2 / 19.7 us per step with, 17.4 / 22.8 without
    // (scripts/ev_step_bench.py, MARL / EV reward, profiles/r05z_ev_step_bench.log against r05w_ev_step_bench.log).
    // The observation epilogue's (cl_step_observe_f32, 65 536 envs): 9.30 - 9.35 us with, 8.84 - 8.85 without (scripts/step_observe_bench.py).
    // (round 6, the HBM-streaming shape 17 x 1 048 576 now that this kernel runs it: the hint on every load, A/B builds in alternating processes --
    //  115.8 - 118.3 us without against 120.0 - 121.3 with on the fp32 map, 120.6 - 123.9 against 113.6 - 121.7 under the chain: nothing; profiles/r06_stream_ntl_ab.log)
    constexpr bool NTL = NT && CL_LEAN_NT_LOADS && (KPI || FLEX || VEC == 2);
    CL_TRACE_DECL;
    CL_TRACE_ENTRY(0);
    CL_TRACE_CYCLES_ENTRY(4);
    if (live) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (!own[m]) continue;
            const long long off = (long long)bb[m] * a.ld + env0;          // every building has its state rows: no flag test
            pload<VEC, NTL>(s_soc[m], a.state + CLS_B_SOC * plane + off);
            pload<VEC, NTL>(s_eff[m], a.state + CLS_B_EFF * plane + off);
            pload<VEC, NTL>(s_deg[m], a.state + CLS_B_DEGCAP * plane + off);
            if constexpr (F64) {
                pload<VEC, NTL>(s_efl[m], a.state + CLS_B_EFF_LO * plane + off);
                pload<VEC, NTL>(s_dgl[m], a.state + CLS_B_DEGCAP_LO * plane + off);
            }
            if (act_by_bldg) pload<VEC, NTL>(a_es[m], a.actions + (long long)bb[m] * a.act_stride_col + env0);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (!own[m]) continue;
        const int b = bb[m];
        cl::Bp B;
        cl::load_bp<false>(B, a.params + (long long)b * CL_NP);
        cl::Row R;
        cl::load_row<false>(R, a.ts + ((long long)ts_row * a.n_bldg + b) * CL_NF, B.flags);
        // (OBS) the building's column descriptors and the offsets of their affine maps, fetched with the parameters: nothing is left to
        // wait for between the arithmetic and the tile writes
        [[maybe_unused]] int od_n = 0, od_col[CL_OBS_FUSED_PER_BLDG], od_src[CL_OBS_FUSED_PER_BLDG];
        [[maybe_unused]] float od_scale[CL_OBS_FUSED_PER_BLDG], od_base[CL_OBS_FUSED_PER_BLDG];
        if constexpr (OBS) {
            const float* __restrict__ trow = of->table + (size_t)(of->row + (ts_row - a.t)) * of->n_cols;
            const int d0 = of->start[b];
            od_n = of->start[b + 1] - d0;
#pragma unroll
            for (int k = 0; k < CL_OBS_FUSED_PER_BLDG; ++k) {
                const int d = min(d0 + min(k, max(od_n - 1, 0)), CLOB_MAX_DEPS - 1);      // (a building without columns reads a valid, unused entry)
                od_col[k] = of->deps[d].col; od_src[k] = of->deps[d].src; od_scale[k] = of->deps[d].scale;
                od_base[k] = trow[od_col[k]];
            }
        }
        if (!live) continue;
        const long long off = (long long)b * a.ld + env0;
        const bool batt = B.flags & CLF_BATTERY;
        if (!act_by_bldg) load_action<VEC>(a_es[m], a, B.a_es, env0);
        CL_TRACE_WAITV(1 + 4 * m, s_deg[m][0]);
        float o_net[VEC], o_rw[VEC];
        [[maybe_unused]] float o_cb[VEC];
        // chargers / washing machines of this building (cl_flex_kernel ran just before this launch)
        const int fbi = (FLEX && (B.flags & CLF_FLEX)) ? (int)B.p[CLP_FLEX_INDEX] : -1;
        float x_load[VEC];
        if (FLEX && fbi >= 0) vload<VEC>(x_load, a.flex_out + (long long)CLX_LOAD * a.n_flex_bldg * a.n_env + (long long)fbi * a.n_env + env0);
        if constexpr (FLEX) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                cl::State S;
                S.soc = batt ? s_soc[m][i] : 0.0f; S.eff = batt ? s_eff[m][i] : 1.0f; S.degcap = batt ? s_deg[m][i] : 0.0f;
                S.cs = S.hs = S.ds = 0.0f;
                const cl::Act act = {0.0f, 0.0f, 0.0f, B.a_es >= 0 ? a_es[m][i] : 0.0f, 0.0f, 0.0f};
                cl::Out O;
                cl::unit_step<false>(B, R, a.t, quirk, act, S, O);
                if (FLEX && fbi >= 0) cl::apply_flex(false, R.price, R.carbon, x_load[i], 0.0f, O);
                const float rw = cl::unit_reward<false>(rkind, B, S, O.net);
                s_soc[m][i] = S.soc; s_eff[m][i] = S.eff; s_deg[m][i] = S.degcap;
                o_net[i] = O.net; o_rw[i] = rw;
                q_net[i] += O.net; q_cost[i] += O.cost; q_em[i] += O.emission;
                q_rw[i] += (FLEX && rkind == CLR_EV) ? cl::marl_reward(O.net, 1.0f) : rw;
            }
        } else {
            // The lean unit (cl::unit_step<false> + cl::unit_reward<false>, restated with the same expressions) with everything that
            // does not depend on the env taken out of the env loop: the t = 0 booking of the load, the battery's curve parameters
            // (in VGPRs once per building instead of one v_mov per select per env), the has-battery and reward-kind tests.
            // 111 -> 90 VALU instructions per unit (SQ_INSTS_VALU).
            const bool first = quirk && a.t == 0;
            float c_ns = first ? 3.0f * R.nsl : R.nsl;           // booked at reset, by the step and by update_variables
            float sol = R.sol;
            const float cbk = first ? 2.0f : 1.0f;               // c_b = first ? 2 eb : eb
            CL_PIN_V(c_ns); CL_PIN_V(sol);
            float soc_rw[VEC];
            if (F64 && batt) {
                // the battery map in float64 on the unrounded parameters (CLD_F64_MAPS, cl_unit.h)
                if constexpr (F64) {
                    cl::BattP64 B64;
                    cl::load_batt64(B64, B.p);
                    if (B.a_es < 0) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) a_es[m][i] = 0.0f;
                    }
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        cl::State S = {s_soc[m][i], s_eff[m][i], s_deg[m][i], 0.0f, 0.0f, 0.0f, s_efl[m][i], s_dgl[m][i]};
                        const float eb = cl::battery_charge_ref(B64, (double)a_es[m][i] * B64.pow * B64.dt / B64.r, a.t == 0, S);     // (flexibility = +inf)
                        s_soc[m][i] = S.soc; s_eff[m][i] = S.eff; s_deg[m][i] = S.degcap; s_efl[m][i] = S.eff_lo; s_dgl[m][i] = S.deg_lo;
                        soc_rw[i] = S.soc;
                        o_cb[i] = cbk * eb;
                        o_net[i] = fmaf(c_ns + o_cb[i], B.r, sol);
                    }
                }
            } else if (PREC == 2 && batt) {
                // the soc chain in float64, the degraded capacity as the capacity loss (CLD_F64_CHAIN, cl_unit.h)
                if constexpr (PREC == 2) {
                    cl::BattC bc;
                    cl::load_battc(bc, B.p);
                    if (B.a_es < 0) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) a_es[m][i] = 0.0f;
                    }
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        cl::State S = {s_soc[m][i], s_eff[m][i], s_deg[m][i], 0.0f, 0.0f, 0.0f};
                        const float eb = cl::battery_charge_chain(bc, a_es[m][i], INFINITY, S);
                        s_soc[m][i] = S.soc; s_eff[m][i] = S.eff; s_deg[m][i] = S.degcap;
                        soc_rw[i] = S.soc;
                        o_cb[i] = cbk * eb;
                        o_net[i] = fmaf(c_ns + o_cb[i], B.r, sol);
                    }
                }
            } else if (batt) {
                cl::BattP Bv = B.batt;
                CL_PIN_V(Bv.cpc_a0); CL_PIN_V(Bv.cpc_b0); CL_PIN_V(Bv.cpc_a1); CL_PIN_V(Bv.cpc_b1);
                CL_PIN_V(Bv.pec_a0); CL_PIN_V(Bv.pec_b0); CL_PIN_V(Bv.pec_a1); CL_PIN_V(Bv.pec_b1);
                CL_PIN_V(Bv.pec_a2); CL_PIN_V(Bv.pec_b2); CL_PIN_V(Bv.pec_a3); CL_PIN_V(Bv.pec_b3);
                if (B.a_es < 0) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) a_es[m][i] = 0.0f;
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    cl::State S = {s_soc[m][i], s_eff[m][i], s_deg[m][i], 0.0f, 0.0f, 0.0f};
                    // battery_step with flexibility = +inf (no outage in a lean district)
                    const float eb = cl::battery_energy(Bv, a_es[m][i] * Bv.pdt, S);
                    s_soc[m][i] = S.soc; s_eff[m][i] = S.eff; s_deg[m][i] = S.degcap;
                    soc_rw[i] = S.soc;
                    o_cb[i] = cbk * eb;                           // the battery's booked consumption (doubled at t = 0)
                    o_net[i] = fmaf(c_ns + o_cb[i], B.r, sol);
                }
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) { o_net[i] = fmaf(c_ns + 0.0f, B.r, sol); soc_rw[i] = 0.0f; o_cb[i] = 0.0f; }
            }
            cl::lean_rewards<VEC>(rkind, B, soc_rw, o_net, o_rw);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                q_net[i] += o_net[i]; q_cost[i] += cl::mul_rn(o_net[i], R.price); q_em[i] += fmaxf(0.0f, o_net[i] * R.carbon);
                q_rw[i] += o_rw[i];
            }
        }
        CL_TRACE_AFTER(2 + 4 * m, o_rw[VEC - 1]);
        if (batt) {
            pstore<VEC, NT>(a.state + CLS_B_SOC * plane + off, s_soc[m]);
            pstore<VEC, NT>(a.state + CLS_B_EFF * plane + off, s_eff[m]);
            pstore<VEC, NT>(a.state + CLS_B_DEGCAP * plane + off, s_deg[m]);
            if constexpr (F64) {
                pstore<VEC, NT>(a.state + CLS_B_EFF_LO * plane + off, s_efl[m]);
                pstore<VEC, NT>(a.state + CLS_B_DEGCAP_LO * plane + off, s_dgl[m]);
            }
        }
        pstore<VEC, NT>(a.out_bldg + CLO_NET * plane + off, o_net);
        if (rkind != CLR_MARL && !(FLEX && rkind == CLR_EV)) pstore<VEC, NT>(a.out_bldg + CLO_REWARD * plane + off, o_rw);
        if constexpr (KPI && !FLEX) {
            // The streaming KPI accumulators of this building, updated by the wave that holds its step in registers (cl_kpi_bldg_kernel's
            // arithmetic, citylearn.py:1136-1323).  A lean district has no outage and serves its whole load, so the unserved-energy sums
            // do not move; and its baseline -- the net without the battery (building.py:345-366) = load + solar -- and its expected
            // energy do not depend on the env at all: those five sums are kept ONCE per block of CL_ROW0_BLOCK envs (the envs that share
            // a table row), at the block's first env.  Per (env, building) only the four control sums move: 32 B on top of the step's 37.
            // (the four loads are issued together, before the first store.  Fetching them at the top of the kernel with the state planes was
            //  measured and is slower -- 17.8 vs 15.9 us at 17 x 65 536: 117 instead of 89 VGPRs and 32 more floats per lane in the first burst)
            float k_pos[VEC], k_net[VEC], k_em[VEC], k_cost[VEC];
            float* kp = a.kpi_bldg + off;
            vload<VEC>(k_pos, kp + (long long)CLK_C_POS * plane); vload<VEC>(k_net, kp + (long long)CLK_C_NET * plane);
            vload<VEC>(k_em, kp + (long long)CLK_C_EMISSION * plane); vload<VEC>(k_cost, kp + (long long)CLK_C_COST * plane);
            // ... and, in the same burst, the block's five env-independent sums (lane 0 of the block's first workgroup; they were five
            // `+=` statements behind the stores below: five dependent round trips on every wave's way to the reduction)
            const bool block_sums = lane == 0 && (blockIdx.x * TILE) % CL_ROW0_BLOCK == 0;
            float b_pos = 0.0f, b_net = 0.0f, b_em = 0.0f, b_cost = 0.0f, b_exp = 0.0f;
            if (block_sums) {
                b_pos = kp[(long long)CLK_B_POS * plane]; b_net = kp[(long long)CLK_B_NET * plane];
                b_em = kp[(long long)CLK_B_EMISSION * plane]; b_cost = kp[(long long)CLK_B_COST * plane];
                b_exp = kp[(long long)CLK_EXPECTED_ALL * plane];
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                k_pos[i] += fmaxf(o_net[i], 0.0f); k_net[i] += o_net[i];
                k_em[i] += fmaxf(o_net[i] * R.carbon, 0.0f); k_cost[i] += fmaxf(o_net[i] * R.price, 0.0f);
            }
            pstore<VEC, NT>(kp + (long long)CLK_C_POS * plane, k_pos); pstore<VEC, NT>(kp + (long long)CLK_C_NET * plane, k_net);
            pstore<VEC, NT>(kp + (long long)CLK_C_EMISSION * plane, k_em); pstore<VEC, NT>(kp + (long long)CLK_C_COST * plane, k_cost);
            const float c_ns0 = (quirk && a.t == 0) ? 3.0f * R.nsl : R.nsl;
            const float base = fmaf(c_ns0, B.r, R.sol);                          // the step's net with the battery term left out
            if (lane == 0) lds[(size_t)a.nw * NQ * TILE + b] = base;               // for the district baseline series (below)
            if (block_sums) {                                                      // (lane 0: kp = this building's row at the block's first env)
                kp[(long long)CLK_B_POS * plane] = b_pos + fmaxf(base, 0.0f);
                kp[(long long)CLK_B_NET * plane] = b_net + base;
                kp[(long long)CLK_B_EMISSION * plane] = b_em + fmaxf(base * R.carbon, 0.0f);
                kp[(long long)CLK_B_COST * plane] = b_cost + fmaxf(base * R.price, 0.0f);
                kp[(long long)CLK_EXPECTED_ALL * plane] = b_exp + R.nsl;
            }
        }
        if constexpr (OBS) {
            // this building's observation columns, into the workgroup's [64 * VEC envs][pitch] tile (streamed out below)
            float* tile = lds + (size_t)a.nw * NQ * TILE;
#pragma unroll
            for (int k = 0; k < CL_OBS_FUSED_PER_BLDG; ++k) {
                if (k >= od_n) break;                                                // wave-uniform
                const int col = od_col[k], src = od_src[k];
                const float scale = od_scale[k], base = od_base[k];
                const bool is_out = (src >> 28) == CLOB_KIND_OUT;
                const int pl = (src >> 20) & 0xFF;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float v = is_out ? (pl == CLO_NET ? o_net[i] : o_rw[i])
                                           : (pl == CLS_B_SOC ? s_soc[m][i] : pl == CLS_B_EFF ? s_eff[m][i] : s_deg[m][i]);
                    tile[(size_t)(lane * VEC + i) * of->pitch + col] = fmaf(v, scale, base);
                }
            }
        }
        CL_TRACE_AFTER(3 + 4 * m, q_rw[0]);
    }
    if constexpr (OBS) {
        if (w == 0) {                                                              // pad columns: zeros (like cl_observe_f32)
            float* tile = lds + (size_t)a.nw * NQ * TILE;
            for (int c = of->n_cols; c < of->pitch; ++c) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) tile[(size_t)(lane * VEC + i) * of->pitch + c] = 0.0f;
            }
        }
    }
    CL_TRACE_AFTER(13, q_net[0]);
    [[maybe_unused]] KpiSeries base_pre;
    [[maybe_unused]] const bool base_writer = KPI && !FLEX && threadIdx.x == blockDim.x - 1 && (blockIdx.x * TILE) % CL_ROW0_BLOCK == 0;
    if constexpr (KPI && !FLEX) {
        // (the baseline series' accumulators: fetched before the reduction's barriers, by the workgroup's LAST thread -- the first ones
        //  carry the control series)
        if (base_writer) kpi_series_fetch(base_pre, a.kpi_env + (long long)CLKE_PER_COND * a.n_env + blockIdx.x * TILE, a.n_env);
    }
    district_reduce<VEC, FLEX, false, KPI && !FLEX>(a, lds, w, lane, env0, live, plane, rkind, q_net, q_cost, q_em, q_rw, a.nw);
    if constexpr (KPI && !FLEX) {
        // baseline condition of the district series: the sum of the buildings' baselines in building order, once per env block
        // (district_reduce's barriers came after every wave's write of its buildings' values)
        if (base_writer) {
            float sum = 0.0f;
            for (int b = 0; b < a.n_bldg; ++b) sum += lds[(size_t)a.nw * NQ * TILE + b];
            kpi_series_apply(a.kpi_env + (long long)CLKE_PER_COND * a.n_env + blockIdx.x * TILE, a.n_env, a.t, sum, base_pre);
        }
    }
    if constexpr (OBS) {
        // (district_reduce's first barrier came after every wave's tile writes.)  The tile's rows are consecutive envs: one contiguous
        // block of the observation matrix, streamed out in 16-byte stores.
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int tile_env0 = blockIdx.x * TILE;
        const int rows = min(TILE, a.n_env - tile_env0);
        const f4* src4 = reinterpret_cast<const f4*>(lds + (size_t)a.nw * NQ * TILE);
        f4* dst4 = reinterpret_cast<f4*>(of->obs + (size_t)tile_env0 * of->pitch);
        const int total4 = rows * of->pitch / 4;
        for (int q = threadIdx.x; q < total4; q += blockDim.x) {
            if constexpr (NT) __builtin_nontemporal_store(src4[q], dst4 + q);
            else dst4[q] = src4[q];
        }
    }
    CL_TRACE_FLUSH();
}

template <int VEC, bool FLEX, bool NT>
__global__ void __launch_bounds__(1024) cl_step_lean_kernel(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lean_step_body<VEC, FLEX, NT, false>(a, lds, nullptr);
}

// CLD_F64_MAPS: the lean step with the battery map in the reference's precision model (cl::battery_charge_ref) -- same launch shape, two more state planes
template <int VEC, bool NT>
__global__ void __launch_bounds__(1024) cl_step_lean_f64_kernel(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lean_step_body<VEC, false, NT, false, false, 1>(a, lds, nullptr);
}

// CLD_F64_CHAIN: the lean step with the soc chain in float64 (cl::battery_charge_chain) -- same launch shape, same three state planes
template <int VEC, bool NT>
__global__ void __launch_bounds__(1024) cl_step_lean_chain_kernel(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lean_step_body<VEC, false, NT, false, false, 2>(a, lds, nullptr);
}

// ... with the streaming KPI accumulators updated by the step launch (cl_step_lean_kpi_kernel's epilogue), and with the compact observation of
// the next row written by it (cl_step_lean_obs_kernel's): the chain changes the battery map, nothing around it
template <int VEC, bool NT>
__global__ void __launch_bounds__(1024) cl_step_lean_kpi_chain_kernel(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lean_step_body<VEC, false, NT, false, true, 2>(a, lds, nullptr);
}

template <int VEC, bool NT>
__global__ void __launch_bounds__(1024) cl_step_lean_obs_chain_kernel(const StepArgs a, const ObsFusedArgs of) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lean_step_body<VEC, false, NT, true, false, 2>(a, lds, &of);
}

template <int VEC, bool NT>
__global__ void __launch_bounds__(1024) cl_step_lean_kpi_kernel(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lean_step_body<VEC, false, NT, false, true>(a, lds, nullptr);
}

template <int VEC, bool NT>
__global__ void __launch_bounds__(1024) cl_step_lean_obs_kernel(const StepArgs a, const ObsFusedArgs of) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lean_step_body<VEC, false, NT, true>(a, lds, &of);
}

// Lean districts, one wave = 64 envs x ALL buildings ("env-major").  The wave issues every state / action load of its
// envs -- 4 dwords per building -- before the first parameter read, then walks the buildings in order while the data streams
// in, storing as it goes: no LDS, no barrier, no cross-wave reduction (the district sums are formed in registers, in building
// order -- the reference's own summation order, citylearn.py:1909-1918).  With one such wave per SIMD the read stream, the
// arithmetic and the write stream of a CU overlap, which the building-major kernels (one generation of waves in lockstep:
// load, then compute, then store) cannot do.  NB = compile-time bound on the buildings held in flight.
// VEC (round 5): envs per lane.  At one env per lane every plane access is a dword per lane -- 256 B per wave instruction; two envs per
// lane (8-byte accesses, 128-thread workgroups so that the env tile still divides CL_ROW0_BLOCK) halve the number of memory instructions
// and of wave-uniform operations per unit, at twice the registers per wave (two waves per SIMD instead of four, the same bytes in flight).
// The district net of building b is parked in the register that held its state of charge (dead by then): no second array for MARL.
// (Round 5, measured and dropped: the action columns through direct-to-LDS loads -- global_load_lds_dword, no vector register holds them
//  while the state planes stream in: 96 instead of 110 registers, five waves per SIMD instead of four -- 17 x 1 048 576: 127.3 - 129.1 us
//  against 125.2 - 125.4 us, alternating on one box (profiles/r05d_*): every load of the wave then has to land before its first building
//  starts, and a fifth wave per SIMD does not pay for that.)
template <int NB, bool NT, int VEC = 1, int PREC = 0>
__global__ void __launch_bounds__(256 / VEC) cl_step_envmajor_kernel(const StepArgs a) {
    constexpr int THREADS = 256 / VEC, TILE = 256;
    const int env = blockIdx.x * TILE + threadIdx.x * VEC;
    const bool live = env < a.n_env;                  // n_env % 4 == 0 (host): a lane's envs are all live or all dead
    const long long plane = (long long)a.n_bldg * a.ld;
    const int rkind = (a.flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT;
    const bool quirk = a.flags & CLD_REF_T0_QUIRK;
    const bool act_by_bldg = (a.flags & CLD_ES_COL_IS_BLDG) && a.act_stride_env == 1;
    const int ts_row = a.t + (a.env_row0 ? a.env_row0[(blockIdx.x * TILE) / CL_ROW0_BLOCK] : 0);
    // the buildings' parameter blocks and time-series rows, staged once per workgroup: a wave walking 17 buildings cannot
    // afford a scalar-load round trip per building (16 us at any batch size), and 17 x 36 SGPRs do not exist
    // (PREC == 2, CLD_F64_CHAIN: the chain's float64 constants -- 2 x CLPC_USED words of the CLP_C_* block -- ride behind them)
    constexpr int PW = CLP_L_LAST - CLP_L_FIRST + 1;          // 32 parameter words
    constexpr int CW = PREC == 2 ? 2 * CLPC_USED : 0;
    __shared__ __attribute__((aligned(8))) uint32_t sp[NB][PW + 4 + CW];
    for (int i = threadIdx.x; i < a.n_bldg * (PW + 4 + CW); i += THREADS) {
        const int b = i / (PW + 4 + CW), k = i - b * (PW + 4 + CW);
        uint32_t v;
        if (k < PW) v = a.params[(long long)b * CL_NP + CLP_L_FIRST + k];
        else if (k >= PW + 4) v = a.params[(long long)b * CL_NP + CLP_C_FIRST + (k - (PW + 4))];
        else {
            const float* q = a.ts + ((long long)ts_row * a.n_bldg + b) * CL_NF;
            v = __float_as_uint(k == PW ? q[CLT_NSL] : k == PW + 1 ? q[CLT_SOLAR] : k == PW + 2 ? q[CLT_PRICE] : q[CLT_CARBON]);
        }
        sp[b][k] = v;
    }
    float s_soc[NB][VEC], s_eff[NB][VEC], s_deg[NB][VEC], a_es[NB][VEC];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s_soc[b][i] = s_eff[b][i] = s_deg[b][i] = a_es[b][i] = 0.0f;
        if (live && b < a.n_bldg) {               // wave-uniform in b; no `break`: it would push the arrays to scratch
            const long long off = (long long)b * a.ld + env;
            // (non-temporal loads here: no effect between 131 072 and 1 048 576 envs, scripts/stream_floor.py -- the copy-floor pattern
            //  gains 9 % from them at 1 048 576 envs, 120 -> 109 us; this kernel's 4-byte-per-lane loads do not)
            vload<VEC>(s_soc[b], a.state + CLS_B_SOC * plane + off);
            vload<VEC>(s_eff[b], a.state + CLS_B_EFF * plane + off);
            vload<VEC>(s_deg[b], a.state + CLS_B_DEGCAP * plane + off);
            if (act_by_bldg) vload<VEC>(a_es[b], a.actions + (long long)b * a.act_stride_col + env);
        }
    }
    __syncthreads();
    if (!live) return;
    float q_net[VEC], q_cost[VEC], q_em[VEC], q_rw[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) q_net[i] = q_cost[i] = q_em[i] = q_rw[i] = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b >= a.n_bldg) continue;
        cl::Bp B;
        cl::load_bp<false>(B, &sp[b][0] - CLP_L_FIRST);           // the CLP_L_* block, from LDS (uniform address: a broadcast read)
        cl::Row R;
        R.nsl = __uint_as_float(sp[b][PW]); R.sol = __uint_as_float(sp[b][PW + 1]);
        R.price = __uint_as_float(sp[b][PW + 2]); R.carbon = __uint_as_float(sp[b][PW + 3]);
        R.outage = false;
        const long long off = (long long)b * a.ld + env;
        const bool batt = B.flags & CLF_BATTERY;
        float o_soc[VEC], o_eff[VEC], o_deg[VEC], o_net[VEC], o_rw[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float act_v = 0.0f;
            if (B.a_es >= 0) act_v = act_by_bldg ? a_es[b][i] : a.actions[(long long)B.a_es * a.act_stride_col + (long long)(env + i) * a.act_stride_env];
            cl::State S;
            S.soc = batt ? s_soc[b][i] : 0.0f; S.eff = batt ? s_eff[b][i] : 1.0f; S.degcap = batt ? s_deg[b][i] : 0.0f;
            S.cs = S.hs = S.ds = 0.0f;
            const cl::Act act = {0.0f, 0.0f, 0.0f, act_v, 0.0f, 0.0f};
            cl::Out O;
            if constexpr (PREC == 2) {
                // the lean unit around cl::battery_charge_chain (same lines as cl::unit_step<false, 2>), its constants read from the LDS copy
                float eb = 0.0f;
                if (batt) {
                    cl::BattC bc;
                    cl::load_battc64(bc, &sp[b][0] + (PW + 4) - CLP_C_FIRST);
                    bc.cap32 = B.batt.cap; bc.omd32 = B.batt.omd; bc.degk = B.batt.degk;       // (the CLP_L_* words: already here)
                    eb = cl::battery_charge_chain(bc, act_v, INFINITY, S);
                }
                const bool first = quirk && a.t == 0;
                const float c_ns = first ? 3.0f * R.nsl : R.nsl, c_b = first ? 2.0f * eb : eb;
                O.net = fmaf(c_ns + c_b, B.r, R.sol); O.cost = cl::mul_rn(O.net, R.price); O.emission = fmaxf(0.0f, O.net * R.carbon);
            } else cl::unit_step<false>(B, R, a.t, quirk, act, S, O);
            o_soc[i] = S.soc; o_eff[i] = S.eff; o_deg[i] = S.degcap; o_net[i] = O.net;
            o_rw[i] = cl::unit_reward<false>(rkind, B, S, O.net);
            s_soc[b][i] = O.net;                                  // (MARL's second sweep reads the nets from here)
            q_net[i] += O.net; q_cost[i] += O.cost; q_em[i] += O.emission; q_rw[i] += o_rw[i];
        }
        if (batt) {
            pstore<VEC, NT>(a.state + CLS_B_SOC * plane + off, o_soc);
            pstore<VEC, NT>(a.state + CLS_B_EFF * plane + off, o_eff);
            pstore<VEC, NT>(a.state + CLS_B_DEGCAP * plane + off, o_deg);
        }
        pstore<VEC, NT>(a.out_bldg + CLO_NET * plane + off, o_net);
        if (rkind != CLR_MARL) pstore<VEC, NT>(a.out_bldg + CLO_REWARD * plane + off, o_rw);
    }
    if (rkind == CLR_MARL) {                      // reward_function.py:132-143: every building against the district net
#pragma unroll
        for (int i = 0; i < VEC; ++i) q_rw[i] = 0.0f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b >= a.n_bldg) continue;
            float rw[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) { rw[i] = cl::marl_reward(s_soc[b][i], q_net[i]); q_rw[i] += rw[i]; }
            vstore<VEC>(a.out_bldg + CLO_REWARD * plane + (long long)b * a.ld + env, rw);
        }
    }
    vstore<VEC>(a.out_env + (long long)CLQ_NET * a.n_env + env, q_net);
    vstore<VEC>(a.out_env + (long long)CLQ_COST * a.n_env + env, q_cost);
    vstore<VEC>(a.out_env + (long long)CLQ_EMISSION * a.n_env + env, q_em);
    vstore<VEC>(a.out_env + (long long)CLQ_REWARD * a.n_env + env, q_rw);
}

// (Round 6, measured and dropped -- two attempts at the HBM-streaming shape, 17 x 1 048 576 envs, profiles/r06d_streaming_ab.log:
//  (1) a PERSISTENT, software-pipelined form: 768 workgroups walking several env tiles each, the registers of building b re-filled with the next tile's
//  planes as soon as building b of the current tile is computed, on the theory that the plain kernel's four resident generations run in lockstep -- a
//  164 MB load burst, then arithmetic with the read path idle, four times over.  The 68 loop-carried plane registers + the unit's temporaries need 168
//  registers (three waves per SIMD, still 112 / 160 bytes of scratch in the fp32 / chain instantiation), the compiler's s_waitcnt placement needs a
//  static building count and a branch-free tile loop to keep the prefetches in flight at all, and the result is 155 - 178 us (fp32) / 228 - 271 us
//  (chain) against 124 - 126 / 134 us: a wave that issues four loads per building paces its memory requests by its own arithmetic.
//  (2) narrower workgroups (one or two waves instead of four, so that waves retire and start one at a time): 142 / 158 us against 134 us.
//  What this shape wants is what it has: every load of a wave issued up front, sixteen waves per CU.)
// (Round 5, measured and dropped -- a building-major kernel for the streaming regime: four-wave workgroups over 256 envs at four envs per lane,
//  wave w taking buildings w, w + 4, ..., all plane loads up front like above but 1 KB per wave instruction instead of 256 B, parameters from an
//  LDS copy, 148 VGPRs = three waves per SIMD.  17 x 1 048 576, alternating runs on one box (profiles/r05g_*): 114.9 / 129.1 / 129.2 us against
//  115.9 / 120.4 / 125.8 us for this kernel -- the same range; each PROCESS lands somewhere in it and stays there for thousands of launches (not a
//  function of the planes' virtual addresses: profiles/r05i_placement.log).  17 x 262 144: 30.9 vs 26.0 us.  Wider accesses are not what this shape lacks.)
#ifndef CL_TU_NOSLP      /* (cl_noslp_tu.hip compiles only what its two kernels need) */
// Second pass for building-chunked launches: add the per-chunk partial district sums.  One workgroup = 64 envs x ONE district
// quantity x 16 waves; wave w adds chunks w, w+16, ... (independent loads issued four at a time: one memory round trip for up
// to 64 chunks), then the 16 wave partials are summed in a fixed order through LDS -- deterministic.  (The first version let one
// workgroup walk all four quantities of its 64 envs: 16 workgroups and four dependent rounds at 1024 envs, 4.7 us of a 22 us step.)
// `deferred` (cl_finish_f32): fold buffer t & 1 of the double-buffered scratch rows, and only if its marker says it holds step t's sums.
// `ret_env` (chunked fused rollout, cl_rollout_f32): workgroup row NQ adds the chunks' shares of the K-step return -- one scratch row per
// chunk behind the n_chunks x NQ partial sums -- to ret_env, in the same association.
__global__ void __launch_bounds__(1024) cl_finish_kernel(const StepArgs a, const int deferred, float* __restrict__ ret_env) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int q = blockIdx.y;
    const long long plane = (long long)a.n_bldg * a.n_env;
    const float* scratch = a.out_bldg + (long long)CLO_RESERVED * plane;
    int n_chunks = a.n_chunks;
    if (!deferred && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)         // (this step did not defer: see district_reduce)
        reinterpret_cast<unsigned*>(a.out_bldg + (long long)(CLO_RESERVED + 1) * plane - 4)[a.t & 1] = 0u;
    if (deferred) {
        const unsigned* marker = reinterpret_cast<const unsigned*>(a.out_bldg + (long long)(CLO_RESERVED + 1) * plane - 4);
        if (marker[a.t & 1] != (unsigned)a.t + 1u) return;                   // (uniform) step t did not defer its finish: out_env is already final
        n_chunks = (int)marker[2];
        scratch += (long long)(a.t & 1) * n_chunks * NQ * a.n_env;
    }
    const int rkind = (a.flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT;
    const bool marl = rkind == CLR_MARL && q == CLQ_REWARD;      // partials carried sign(-net) * 0.01 * net^2: scale by max(0, district net)
    float s = 0.0f, sn = 0.0f;
    if (e < a.n_env) {
        for (int c0 = w; c0 < n_chunks; c0 += 64) {
            float v[4], vn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + 16 * j;
                v[j] = c < n_chunks ? scratch[(q < NQ ? (long long)c * NQ + q : (long long)n_chunks * NQ + c) * a.n_env + e] : 0.0f;
                vn[j] = (marl && c < n_chunks) ? scratch[((long long)c * NQ + CLQ_NET) * a.n_env + e] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { s += v[j]; sn += vn[j]; }
        }
    }
    part[w][lane] = s;
    __syncthreads();
    float t = 0.0f;
    if (w == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][lane];
    }
    if (marl) {                                  // the district net, summed in the same order as the CLQ_NET workgroup does
        __syncthreads();
        part[w][lane] = sn;
        __syncthreads();
        if (w == 0) {
            float n = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) n += part[k][lane];
            t *= fmaxf(0.0f, n);
        }
    }
    if (w == 0 && e < a.n_env) {
        if (q < NQ) a.out_env[(long long)q * a.n_env + e] = t;
        else ret_env[e] += t;
    }
}

// ---- streaming KPI accumulators (CLD_KPI): two small passes over the detail planes the step kernel just wrote ----
__global__ void cl_kpi_bldg_kernel(const StepArgs a) {
    const long long plane = (long long)a.n_bldg * a.n_env;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    const int b = (int)(i / a.n_env);
    const int ts_row = a.t + (a.env_row0 ? a.env_row0[(int)(i - (long long)b * a.n_env) / CL_ROW0_BLOCK] : 0);
    const float* q = a.ts + ((long long)ts_row * a.n_bldg + b) * CL_NF;
    const float price = q[CLT_PRICE], carbon = q[CLT_CARBON];
    const bool outage = q[CLT_OUTAGE] != 0.0f;
    const float net = a.out_bldg[CLO_NET * plane + i], base = a.out_bldg[CLO_BASE_NET * plane + i];
    const float ex = a.out_bldg[CLO_EXPECTED * plane + i], sv = a.out_bldg[CLO_SERVED * plane + i];
    float* k = a.kpi_bldg + i;
    k[CLK_C_POS * plane] += fmaxf(net, 0.0f);
    k[CLK_C_NET * plane] += net;
    k[CLK_C_EMISSION * plane] += fmaxf(net * carbon, 0.0f);
    k[CLK_C_COST * plane] += fmaxf(net * price, 0.0f);
    k[CLK_B_POS * plane] += fmaxf(base, 0.0f);
    k[CLK_B_NET * plane] += base;
    k[CLK_B_EMISSION * plane] += fmaxf(base * carbon, 0.0f);
    k[CLK_B_COST * plane] += fmaxf(base * price, 0.0f);
    if (outage) { k[CLK_UNSERVED_OUTAGE * plane] += ex - sv; k[CLK_EXPECTED_OUTAGE * plane] += ex; }
    k[CLK_UNSERVED_ALL * plane] += ex - sv;
    k[CLK_EXPECTED_ALL * plane] += ex;
}

CL_DEV void kpi_series_fetch(KpiSeries& s, const float* __restrict__ k, long long n_env) {
    s.prev = k[CLKE_PREV * n_env]; s.ramp = k[CLKE_RAMP * n_env];
    s.dsum = k[CLKE_DAY_SUM * n_env]; s.dmax = k[CLKE_DAY_MAX * n_env];
    s.msum = k[CLKE_MON_SUM * n_env]; s.mmax = k[CLKE_MON_MAX * n_env];
    s.amax = k[CLKE_ALL_MAX * n_env];
}

CL_DEV void kpi_series_apply(float* __restrict__ k, long long n_env, int t, float v, const KpiSeries& s) {
    // one more sample `v` (index t) of a district series: ramping, 24- and 730-step load factor / peak groups
    if (t > 0) k[CLKE_RAMP * n_env] = s.ramp + fmaxf(v - s.prev, 0.0f);
    k[CLKE_PREV * n_env] = v;
    float dsum = s.dsum, dmax = s.dmax;
    if (t > 0 && t % 24 == 0) {
        k[CLKE_DAY_LF_SUM * n_env] += 1.0f - (dsum * (1.0f / 24.0f)) / dmax;
        k[CLKE_DAY_PEAK_SUM * n_env] += dmax;
        k[CLKE_DAY_N * n_env] += 1.0f;
        dsum = 0.0f; dmax = -INFINITY;
    }
    k[CLKE_DAY_SUM * n_env] = dsum + v; k[CLKE_DAY_MAX * n_env] = fmaxf(dmax, v);
    float msum = s.msum, mmax = s.mmax;
    if (t > 0 && t % 730 == 0) {
        k[CLKE_MON_LF_SUM * n_env] += 1.0f - (msum * (1.0f / 730.0f)) / mmax;
        k[CLKE_MON_N * n_env] += 1.0f;
        msum = 0.0f; mmax = -INFINITY;
    }
    k[CLKE_MON_SUM * n_env] = msum + v; k[CLKE_MON_MAX * n_env] = fmaxf(mmax, v);
    k[CLKE_ALL_MAX * n_env] = fmaxf(s.amax, v);
}

CL_DEV void kpi_series_update(float* __restrict__ k, long long n_env, int t, float v) {
    KpiSeries s;
    kpi_series_fetch(s, k, n_env);
    kpi_series_apply(k, n_env, t, v, s);
}

// District series: control = out_env net; baseline = sum over buildings of the baseline plane (16 waves share the
// building loop, fixed-order LDS sum).
__global__ void __launch_bounds__(1024) cl_kpi_env_kernel(const StepArgs a) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const long long plane = (long long)a.n_bldg * a.n_env;
    float s = 0.0f;
    if (e < a.n_env)
        for (int b = w; b < a.n_bldg; b += 16) s += a.out_bldg[CLO_BASE_NET * plane + (long long)b * a.n_env + e];
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < a.n_env) {
        float base = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) base += part[k][lane];
        KpiSeries sc, sb;                       // both series' accumulators first, then the stores (one round trip instead of ten)
        kpi_series_fetch(sc, a.kpi_env + e, a.n_env);
        kpi_series_fetch(sb, a.kpi_env + (long long)CLKE_PER_COND * a.n_env + e, a.n_env);
        const float dnet = a.out_env[(long long)CLQ_NET * a.n_env + e];
        kpi_series_apply(a.kpi_env + e, a.n_env, a.t, dnet, sc);
        kpi_series_apply(a.kpi_env + (long long)CLKE_PER_COND * a.n_env + e, a.n_env, a.t, base, sb);
    }
}

// The two passes above in ONE launch (round 3): a workgroup = 64 envs x every building (16 waves share the building loop); a wave updates
// the per-building accumulators of its buildings from the planes the step just wrote and keeps the baseline partial sum of its envs in a
// register, the 16 partials are added in wave order through LDS (the order cl_kpi_env_kernel used: same bits) and wave 0 feeds the two
// district series.  One launch and one read of the baseline plane less per step.
__global__ void __launch_bounds__(1024) cl_kpi_kernel(const StepArgs a) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int e = blockIdx.x * 64 + lane;
    const bool live = e < a.n_env;
    const long long plane = (long long)a.n_bldg * a.n_env;
    const int ts_row = a.t + (a.env_row0 ? a.env_row0[(blockIdx.x * 64) / CL_ROW0_BLOCK] : 0);      // 64 divides CL_ROW0_BLOCK: workgroup-uniform
    float s = 0.0f;
    for (int b = w; b < a.n_bldg; b += 16) {
        const float* __restrict__ q = a.ts + ((long long)ts_row * a.n_bldg + b) * CL_NF;
        const float price = q[CLT_PRICE], carbon = q[CLT_CARBON];
        const bool outage = q[CLT_OUTAGE] != 0.0f;
        if (!live) continue;
        const long long i = (long long)b * a.n_env + e;
        const float net = a.out_bldg[CLO_NET * plane + i], base = a.out_bldg[CLO_BASE_NET * plane + i];
        const float ex = a.out_bldg[CLO_EXPECTED * plane + i], sv = a.out_bldg[CLO_SERVED * plane + i];
        // every accumulator is LOADED before the first one is stored: `k[p * plane] += x` one after the other is a chain of dependent
        // round trips (a store to k[p * plane] may alias the next load as far as the compiler knows) -- that chain, not the bytes, was
        // the 20 us of the round-2 passes
        float* k = a.kpi_bldg + i;
        float v[CL_NKB];
#pragma unroll
        for (int p = 0; p < CL_NKB; ++p) v[p] = (outage || (p != CLK_UNSERVED_OUTAGE && p != CLK_EXPECTED_OUTAGE)) ? k[p * plane] : 0.0f;
        v[CLK_C_POS] += fmaxf(net, 0.0f);
        v[CLK_C_NET] += net;
        v[CLK_C_EMISSION] += fmaxf(net * carbon, 0.0f);
        v[CLK_C_COST] += fmaxf(net * price, 0.0f);
        v[CLK_B_POS] += fmaxf(base, 0.0f);
        v[CLK_B_NET] += base;
        v[CLK_B_EMISSION] += fmaxf(base * carbon, 0.0f);
        v[CLK_B_COST] += fmaxf(base * price, 0.0f);
        v[CLK_UNSERVED_OUTAGE] += ex - sv; v[CLK_EXPECTED_OUTAGE] += ex;
        v[CLK_UNSERVED_ALL] += ex - sv;
        v[CLK_EXPECTED_ALL] += ex;
#pragma unroll
        for (int p = 0; p < CL_NKB; ++p)
            if (outage || (p != CLK_UNSERVED_OUTAGE && p != CLK_EXPECTED_OUTAGE)) k[p * plane] = v[p];       // (row-uniform: the outage sums are untouched otherwise)
        s += base;
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && live) {
        float base = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) base += part[k][lane];
        KpiSeries sc, sb;                       // both series' accumulators first, then the stores (one round trip instead of ten)
        kpi_series_fetch(sc, a.kpi_env + e, a.n_env);
        kpi_series_fetch(sb, a.kpi_env + (long long)CLKE_PER_COND * a.n_env + e, a.n_env);
        const float dnet = a.out_env[(long long)CLQ_NET * a.n_env + e];
        kpi_series_apply(a.kpi_env + e, a.n_env, a.t, dnet, sc);
        kpi_series_apply(a.kpi_env + (long long)CLKE_PER_COND * a.n_env + e, a.n_env, a.t, base, sb);
    }
}

// Third pass, MARL only: per-building rewards need the finished district net.
__global__ void cl_marl_reward_kernel(const StepArgs a) {
    const long long plane = (long long)a.n_bldg * a.n_env;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    const int e = (int)(i % a.n_env);
    a.out_bldg[CLO_REWARD * plane + i] = cl::marl_reward(a.out_bldg[CLO_NET * plane + i], a.out_env[(long long)CLQ_NET * a.n_env + e]);
}

__global__ void cl_reset_kernel(const uint32_t* __restrict__ params, float* __restrict__ state,
                                float* __restrict__ kpi_bldg, float* __restrict__ kpi_env, int n_env, int n_bldg, uint32_t flags, int ld) {
    // (`ld`: cl_dims.env_pitch -- the pad entries of a row are initialised like its envs; the KPI planes are never pitched: host)
    const long long plane = (long long)n_bldg * ld;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < plane) {
        const int b = (int)(i / ld);
        const uint32_t* p = params + (long long)b * CL_NP;
        if (state) {
            state[CLS_B_SOC * plane + i] = __uint_as_float(p[CLP_L_SOC0]);
            state[CLS_B_EFF * plane + i] = __uint_as_float(p[CLP_L_EFF0]);
            // (CLD_F64_CHAIN: the plane carries the capacity LOSS, capacity - degraded_capacity: nothing lost yet)
            state[CLS_B_DEGCAP * plane + i] = (flags & CLD_F64_CHAIN) ? 0.0f : __uint_as_float(p[CLP_L_CAP]);
            state[CLS_CS_SOC * plane + i] = __uint_as_float(p[CLP_CS_SOC0]);
            state[CLS_HS_SOC * plane + i] = __uint_as_float(p[CLP_HS_SOC0]);
            state[CLS_DS_SOC * plane + i] = __uint_as_float(p[CLP_DS_SOC0]);
            // CLD_F64_MAPS: what the float32 planes above lose of the two float64 start values (Battery.reset, energy_model.py:1237-1242)
            const double eff0 = cl::pd(p, CLPD_EFF0), cap0 = cl::pd(p, CLPD_CAP);
            state[CLS_B_EFF_LO * plane + i] = (float)(eff0 - (double)__uint_as_float(p[CLP_L_EFF0]));
            state[CLS_B_DEGCAP_LO * plane + i] = (float)(cap0 - (double)__uint_as_float(p[CLP_L_CAP]));
        }
        if (kpi_bldg)
            for (int k = 0; k < CL_NKB; ++k) kpi_bldg[k * plane + i] = 0.0f;
    }
    if (kpi_env && i < n_env)
        for (int k = 0; k < CL_NKE; ++k) {
            const int kk = k % CLKE_PER_COND;
            const bool is_max = kk == CLKE_DAY_MAX || kk == CLKE_MON_MAX || kk == CLKE_ALL_MAX;
            kpi_env[(long long)k * n_env + i] = is_max ? -INFINITY : 0.0f;
        }
}
#endif  // CL_TU_NOSLP

}  // namespace

#ifndef CL_TU_NOSLP
#include "cl_full.h"
#endif
#include "cl_rollout.h"

// Two kernels live in a translation unit of their own (cl_noslp_tu.hip = this file with CL_TU_NOSLP, built with -fno-slp-vectorize):
// with several envs per lane the SLP vectoriser packs the envs' identical fp32 operations into v_pk_*_f32, which cost more than the
// plain operations they replace.  Measured with both builds alternating on one box (profiles/r02e_noslp_ab2.log, r02f_noslp_lean_ab.log):
//   fused rollout   17 x 32 768: 2.34 -> 2.14 us per step, 65 536: 4.26 -> 3.88
//   lean step       17 x 65 536 (the headline launch, four envs per lane, 292 packed operations): 7.24 / 6.92 / 7.24 / 7.28 us with,
//                   7.02 / 6.84 / 6.88 / 6.70 us without
// The same switch is a loss for the LSTM kernel (109.8 -> 115.5 us) and the chunked thermal launches (14.75 -> 15.08 us) and within
// the noise for the env-major kernel, hence per translation unit and not for the whole library.  The lean launches with a fused
// epilogue (flexible loads, KPI accumulators, observation tile) were not measured and stay in the main unit.
extern "C" __attribute__((visibility("hidden"))) int cl_tu_launch_rollout(int key, int pin, unsigned grid, unsigned grid_y, unsigned block, size_t lds,
                                                                          void* stream, const void* rollout_args);
extern "C" __attribute__((visibility("hidden"))) int cl_tu_launch_lean(int vec, int nt, unsigned grid_x, unsigned grid_y, unsigned block, size_t lds,
                                                                       void* stream, const void* step_args);

#ifdef CL_TU_NOSLP
extern "C" __attribute__((visibility("hidden"))) int cl_tu_launch_rollout(int key, int pin, unsigned grid_x, unsigned grid_y, unsigned block_threads, size_t lds,
                                                                          void* stream, const void* rollout_args) {
    const RolloutArgs& r = *static_cast<const RolloutArgs*>(rollout_args);       // the struct of the including translation unit: same source
    const dim3 block(block_threads), grid(grid_x, grid_y);
    hipStream_t s = (hipStream_t)stream;
    switch (key) {        // (full ? 100 : 0) + 10 * envs per lane + buildings per wave
    case 11: hipLaunchKernelGGL((cl_rollout_kernel<1, false, 1>), grid, block, lds, s, r); break;
    case 12:
        if (pin) hipLaunchKernelGGL((cl_rollout_kernel<1, false, 2>), grid, block, lds, s, r);
        else hipLaunchKernelGGL((cl_rollout_kernel<1, false, 2, false>), grid, block, lds, s, r);      // whole launch resident at once: see PIN
        break;
    case 21: hipLaunchKernelGGL((cl_rollout_kernel<2, false, 1>), grid, block, lds, s, r); break;
    case 22: hipLaunchKernelGGL((cl_rollout_kernel<2, false, 2>), grid, block, lds, s, r); break;
    case 111: hipLaunchKernelGGL((cl_rollout_kernel<1, true, 1>), grid, block, lds, s, r); break;
    // CLD_F64_CHAIN (one env per lane: 2000 + key)
    case 2012: hipLaunchKernelGGL((cl_rollout_kernel<1, false, 2, true, false, 2>), grid, block, lds, s, r); break;
    case 2022: hipLaunchKernelGGL((cl_rollout_kernel<2, false, 2, true, false, 2>), grid, block, lds, s, r); break;
    case 2111: hipLaunchKernelGGL((cl_rollout_kernel<1, true, 1, true, false, 2>), grid, block, lds, s, r); break;
    case 3012: hipLaunchKernelGGL((cl_rollout_kernel<1, false, 2, true, true, 2>), grid, block, lds, s, r); break;
    case 3022: hipLaunchKernelGGL((cl_rollout_kernel<2, false, 2, true, true, 2>), grid, block, lds, s, r); break;
    case 3111: hipLaunchKernelGGL((cl_rollout_kernel<1, true, 1, true, true, 2>), grid, block, lds, s, r); break;
    // building-chunked districts (gridDim.y workgroup rows; cl_rollout.h)
    case 1012: hipLaunchKernelGGL((cl_rollout_kernel<1, false, 2, true, true>), grid, block, lds, s, r); break;
    case 1022: hipLaunchKernelGGL((cl_rollout_kernel<2, false, 2, true, true>), grid, block, lds, s, r); break;
    case 1111: hipLaunchKernelGGL((cl_rollout_kernel<1, true, 1, true, true>), grid, block, lds, s, r); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("hidden"))) int cl_tu_launch_lean(int vec, int nt, unsigned grid_x, unsigned grid_y, unsigned block_threads,
                                                                       size_t lds, void* stream, const void* step_args) {
    const StepArgs& a = *static_cast<const StepArgs*>(step_args);
    const dim3 grid(grid_x, grid_y), block(block_threads);
    hipStream_t s = (hipStream_t)stream;
#define CL_TU_LEAN(V) case V: \
        if (nt) hipLaunchKernelGGL((cl_step_lean_kernel<V, false, true>), grid, block, lds, s, a); \
        else hipLaunchKernelGGL((cl_step_lean_kernel<V, false, false>), grid, block, lds, s, a); \
        break;
    switch (vec) {
        CL_TU_LEAN(1) CL_TU_LEAN(2) CL_TU_LEAN(4)
    default: return -1;
    }
#undef CL_TU_LEAN
    return (int)hipGetLastError();
}
#else

#include "cl_lstm.h"
#include "cl_observe.h"

namespace {

int check_dims(const cl_dims* d) {
    if (!d) return fail(CL_ENULL, "dims is NULL");
    if (d->n_env <= 0 || d->n_bldg <= 0 || d->n_steps <= 0 || d->n_act_cols < 0)
        return fail(CL_EINVAL, "bad dims: n_env=%d n_bldg=%d n_steps=%d n_act_cols=%d", d->n_env, d->n_bldg,
                    d->n_steps, d->n_act_cols);
    if (d->n_env % 4 != 0) return fail(CL_EALIGN, "n_env=%d must be a multiple of 4 (pad the env batch)", d->n_env);
    if (d->n_ts_rows != 0 && d->n_ts_rows < d->n_steps)
        return fail(CL_EINVAL, "n_ts_rows=%d < n_steps=%d", d->n_ts_rows, d->n_steps);
    if (reinterpret_cast<uintptr_t>(d->env_row0) & 3) return fail(CL_EALIGN, "env_row0 is not 4-byte aligned");
    if (d->env_pitch != 0 && (d->env_pitch < d->n_env || d->env_pitch % 4 != 0))
        return fail(CL_EINVAL, "env_pitch=%d must be 0 or a multiple of 4 >= n_env=%d", d->env_pitch, d->n_env);
    const uint32_t rk = (d->flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT;
    if (rk > CLR_EV) return fail(CL_EINVAL, "unknown reward kind %u", rk);
    // the Philox counter word of the random streams is env_offset + env (32 bits): shards must not alias
    if (d->env_offset < 0 || d->env_offset + (int64_t)d->n_env > (int64_t)1 << 32)
        return fail(CL_ERANGE, "env_offset=%lld with n_env=%d leaves the 32-bit env index of the random streams", (long long)d->env_offset, d->n_env);
    return CL_OK;
}

// cl_dims.env_pitch (0 = n_env); entry points that do not implement a pitch refuse one
int pitch_of(const cl_dims* d) { return d->env_pitch ? d->env_pitch : d->n_env; }
int no_pitch(const cl_dims* d, const char* who) {
    if (pitch_of(d) != d->n_env) return fail(CL_EINVAL, "%s: env_pitch=%d != n_env=%d is not implemented for this call", who, d->env_pitch, d->n_env);
    return CL_OK;
}

int check_ptr(const void* p, const char* name, bool required = true) {
    if (!p) return required ? fail(CL_ENULL, "%s is NULL", name) : CL_OK;
    if (reinterpret_cast<uintptr_t>(p) & 15) return fail(CL_EALIGN, "%s is not 16-byte aligned", name);
    return CL_OK;
}

// Launch geometry.  Measured on MI355X (scripts/tune.py, profiles/): at ~1M units per launch the 16-wave
// workgroup with 16-byte accesses (1 workgroup per CU, every wave one memory round trip) is fastest; the
// full (thermal) kernel needs too many registers for VEC > 1.
int pick_nw(int n_bldg, int /*vec*/) {
    const int rounds = (n_bldg + 15) / 16;
    int nw = (n_bldg + rounds - 1) / rounds;
    if (n_bldg > 16 && n_bldg <= 32) nw = 16;            // e.g. 17 buildings: 16 waves, wave 0 takes two
    return nw;
}

// envs per lane: wide (16 B) accesses once the batch is large enough to still give every CU a workgroup
int pick_vec(int n_env, int n_bldg, bool unit_stride) {
    if (!unit_stride) return 1;
    // the narrowest pack that still gives at most one workgroup per CU (the lean kernels' launch shape): 49 152 envs at two envs per
    // lane were 384 workgroups -- the general kernel, 9.8 us -- and are 192 latency-ordered ones at four (scripts/step_observe_bench.py)
    if (n_env > 256 * 128) return 4;
    if (n_env > 256 * 64) return 2;
    (void)n_bldg;
    return 1;
}

// Launch-geometry overrides travel with every call (cl_dims.tuning, include/citylearn_amd.h): the library holds no
// mutable state besides the thread-local error string.
const cl_tuning k_default_tuning = {};
// launch K<..., NT> with NT = a.nt (expects grid, block, lds, s, a in scope)
#define CL_LAUNCH_NT(K, ...) do { \
    name_add(tun, #K "<" #__VA_ARGS__ ", %s>", a.nt ? "true" : "false"); \
    if (a.nt) hipLaunchKernelGGL((K<__VA_ARGS__, true>), grid, block, lds, s, a); \
    else hipLaunchKernelGGL((K<__VA_ARGS__, false>), grid, block, lds, s, a); } while (0)

const cl_tuning& tuning_of(const cl_dims* d) { return d->tuning ? *d->tuning : k_default_tuning; }

// cl_tuning.kernel_name (diagnostics): the instantiations a call launched, '+'-separated, spelled as rocprofv3 prints them
void name_reset(const cl_tuning& tun) { if (tun.kernel_name) tun.kernel_name[0] = 0; }
void name_add(const cl_tuning& tun, const char* fmt, ...) {
    if (!tun.kernel_name) return;
    size_t n = strnlen(tun.kernel_name, CL_KERNEL_NAME_LEN - 1);
    if (n && n + 2 < CL_KERNEL_NAME_LEN) { tun.kernel_name[n++] = '+'; tun.kernel_name[n] = 0; }
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tun.kernel_name + n, CL_KERNEL_NAME_LEN - n, fmt, ap);
    va_end(ap);
}

}  // namespace


namespace {
// The rollout policy as planes of actions (same stream as cl_rollout_kernel: a = low + u (high - low),
// u = Philox4x32-10(seed; env, column, t)) -- for districts whose state machines live in HBM between steps.  One Philox
// block holds the draws of four consecutive steps (word t & 3 of block t >> 2), so one launch fills the four planes
// actions[t & 3][column][env] of steps 4 tq .. 4 tq + 3 (one plane per launch measured 19 us at 26 x 65 536: the ten
// rounds of quarter-rate 32-bit multiplies, three of four words thrown away).
__global__ void cl_policy_kernel(float* __restrict__ actions, const float* __restrict__ low, const float* __restrict__ high,
                                 unsigned long long seed, int n_env, int n_cols, int tq, unsigned env_offset) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
    if (env >= n_env) return;
    const float lo = low[col], span = high[col] - lo;
    const cl::U4 blk = cl::philox_block(seed, (uint32_t)env + env_offset, (uint32_t)col, (uint32_t)tq);
#pragma unroll
    for (int w = 0; w < 4; ++w) actions[((long long)w * n_cols + col) * n_env + env] = fmaf(cl::u01(blk.w[w]), span, lo);
}

__global__ void cl_kpi_comfort_reset_kernel(float* k, long long plane) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    for (int p = 0; p < CL_NKC; ++p) k[p * plane + i] = 0.0f;
    k[CLKC_COLD_MIN * plane + i] = INFINITY; k[CLKC_HOT_MIN * plane + i] = INFINITY;
    k[CLKC_COLD_MAX * plane + i] = -INFINITY; k[CLKC_HOT_MAX * plane + i] = -INFINITY;
}

__global__ void cl_return_kernel(float* __restrict__ ret_env, const float* __restrict__ reward, int n_env) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env < n_env) ret_env[env] += reward[env];
}

// the thermal kernels that write the compact observation themselves (cl_full.h OBS): more than 64 KB of dynamic LDS where the tile asks for it
template <int PREC>
int launch_full_obs(bool nt, dim3 grid, dim3 block, size_t lds, hipStream_t s, const StepArgs& a, const ObsFusedArgs& of, const cl_tuning& tun) {
    name_add(tun, "cl_step_full_obs_kernel<%d, %s>", PREC, nt ? "true" : "false");
    const void* fn = nt ? reinterpret_cast<const void*>(cl_step_full_obs_kernel<PREC, true>) : reinterpret_cast<const void*>(cl_step_full_obs_kernel<PREC, false>);
    if (lds > 64 * 1024)
        if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(cl_step_full_obs_kernel)");
    if (nt) hipLaunchKernelGGL((cl_step_full_obs_kernel<PREC, true>), grid, block, lds, s, a, of);
    else hipLaunchKernelGGL((cl_step_full_obs_kernel<PREC, false>), grid, block, lds, s, a, of);
    return CL_OK;
}
template <int VEC, int PREC>
int launch_tp_obs(bool nt, dim3 grid, dim3 block, size_t lds, hipStream_t s, const StepArgs& a, int tp, const ObsFusedArgs& of, const cl_tuning& tun) {
    name_add(tun, "cl_step_full_tp_obs_kernel<%d, %d, %s>", VEC, PREC, nt ? "true" : "false");
    if (lds > 150 * 1024) return fail(CL_EINVAL, "fused observation tile: %zu bytes of LDS", lds);
    const void* fn = nt ? reinterpret_cast<const void*>(cl_step_full_tp_obs_kernel<VEC, PREC, true>) : reinterpret_cast<const void*>(cl_step_full_tp_obs_kernel<VEC, PREC, false>);
    if (lds > 64 * 1024)
        if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(cl_step_full_tp_obs_kernel)");
    if (nt) hipLaunchKernelGGL((cl_step_full_tp_obs_kernel<VEC, PREC, true>), grid, block, lds, s, a, tp, of);
    else hipLaunchKernelGGL((cl_step_full_tp_obs_kernel<VEC, PREC, false>), grid, block, lds, s, a, tp, of);
    return CL_OK;
}

// cl_step_lean_chunk_kernel<VEC, NT, FOLD, PREC> from run-time launch parameters (VEC = 1 or 4); opts into more than 64 KB of dynamic LDS where the launch needs it
template <int PREC>
int launch_lean_chunk(int vec, bool nt, bool fold, dim3 grid, dim3 block, size_t lds, hipStream_t s, const StepArgs& a) {
#define CL_LC(V, N, F) do { \
        if (lds > 64 * 1024) { \
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(cl_step_lean_chunk_kernel<V, N, F, PREC>), lds); e != hipSuccess) \
                return hip_fail(e, "hipFuncSetAttribute(cl_step_lean_chunk_kernel)"); \
        } \
        hipLaunchKernelGGL((cl_step_lean_chunk_kernel<V, N, F, PREC>), grid, block, lds, s, a); } while (0)
#define CL_LC_NF(V) do { if (nt) { if (fold) CL_LC(V, true, true); else CL_LC(V, true, false); } \
                         else { if (fold) CL_LC(V, false, true); else CL_LC(V, false, false); } } while (0)
    if (vec == 1) CL_LC_NF(1); else if (vec == 2) CL_LC_NF(2); else CL_LC_NF(4);
#undef CL_LC_NF
#undef CL_LC
    return CL_OK;
}
}  // namespace

extern "C" {

int cl_abi_version(void) { return CL_ABI_VERSION; }

#ifdef CL_TRACE
// diagnostic build only (scripts/wave_timeline.py): where cl_step_full_kernel parks its per-wave phase stamps
int cl_trace_set(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_cl_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
#endif

const char* cl_last_error(void) { return g_err; }

int cl_reset_f32(const cl_dims* dims, const uint32_t* params, float* state, float* kpi_bldg, float* kpi_env,
                 void* stream) {
    if (int rc = check_dims(dims)) return rc;
    if (int rc = check_ptr(params, "params")) return rc;
    if (int rc = check_ptr(state, "state", false)) return rc;
    if (int rc = check_ptr(kpi_bldg, "kpi_bldg", false)) return rc;
    if (int rc = check_ptr(kpi_env, "kpi_env", false)) return rc;
    const int ld = pitch_of(dims);
    if (ld != dims->n_env && (kpi_bldg || kpi_env)) return fail(CL_EINVAL, "env_pitch=%d: the streaming KPI planes are not pitched", ld);
    const long long n = (long long)ld * dims->n_bldg;
    const int block = 256;
    const unsigned grid = (unsigned)((n + block - 1) / block);
    hipLaunchKernelGGL(cl_reset_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, params, state, kpi_bldg,
                       kpi_env, dims->n_env, dims->n_bldg, dims->flags, ld);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_fail(e, "cl_reset_kernel launch");
    return CL_OK;
}

int cl_step_f32(const cl_dims* dims, const uint32_t* params, const float* ts, float* state, const float* actions,
                int64_t act_stride_col, int64_t act_stride_env, float* out_bldg, float* out_env, float* kpi_bldg,
                float* kpi_env, int32_t t, void* stream) {
    return cl_step_flex_f32(dims, params, ts, state, actions, act_stride_col, act_stride_env, out_bldg, out_env, kpi_bldg,
                            kpi_env, nullptr, t, stream);
}

static int check_flex(const cl_dims* dims, const cl_flex* f) {
    if (f->n_ev < 0 || f->n_flex_bldg <= 0)
        return fail(CL_EINVAL, "cl_flex: bad counts (ev %d, buildings %d)", f->n_ev, f->n_flex_bldg);
    const int rows = dims->n_ts_rows ? dims->n_ts_rows : dims->n_steps;
    if (f->n_rows < rows) return fail(CL_ERANGE, "cl_flex.n_rows=%d but the step tables have %d rows", f->n_rows, rows);
    if (int rc = check_ptr(f->flex_out, "flex.flex_out")) return rc;
    if (int rc = check_ptr(f->ev_params, "flex.ev_params", f->n_ev > 0)) return rc;
    if (int rc = check_ptr(f->ev_ts, "flex.ev_ts", f->n_ev > 0)) return rc;
    if (int rc = check_ptr(f->ev_state, "flex.ev_state", f->n_ev > 0)) return rc;
    if (int rc = check_ptr(f->charger_params, "flex.charger_params")) return rc;
    if (int rc = check_ptr(f->charger_ts, "flex.charger_ts")) return rc;
    if (int rc = check_ptr(f->wm_params, "flex.wm_params")) return rc;
    if (int rc = check_ptr(f->wm_ts, "flex.wm_ts")) return rc;
    if (int rc = check_ptr(f->wm_state, "flex.wm_state")) return rc;
    return CL_OK;
}

int cl_flex_reset_f32(const cl_dims* dims, const cl_flex* flex, void* stream) {
    if (int rc = check_dims(dims)) return rc;
    if (!flex) return fail(CL_ENULL, "flex is NULL");
    if (int rc = no_pitch(dims, "cl_flex_reset_f32")) return rc;
    if (int rc = check_flex(dims, flex)) return rc;
    const int wm_slots = flex->n_flex_bldg * CL_MAXW;
    const long long n = (long long)dims->n_env * (flex->n_ev > wm_slots ? flex->n_ev : wm_slots);
    if (n > 0) {
        hipLaunchKernelGGL(cl_flex_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *flex,
                           dims->env_row0, dims->n_env);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_fail(e, "cl_flex_reset_kernel launch");
    }
    return CL_OK;
}

// `of` / `fused`: the compact observation to write from inside the step launch (cl_step_observe_f32); *fused tells the caller
// whether the launch that ran could take it (lean district, four envs per lane, one workgroup row) or the observation is still to do.

static int step_impl(const cl_dims* dims, const uint32_t* params, const float* ts, float* state, const float* actions,
                     int64_t act_stride_col, int64_t act_stride_env, float* out_bldg, float* out_env, float* kpi_bldg,
                     float* kpi_env, const cl_flex* flex, int32_t t, void* stream, const ObsFusedArgs* of, bool* fused) {
    // (`of`: the compact observation the step launch may write itself -- the battery + PV kernels under their own limits (`lean_ok`), the thermal
    //  kernels of cl_full.h for any listed battery / tank / net / reward column)
    const ObsFusedArgs* const of_full = of;
    if (of && !of->lean_ok) of = nullptr;
    if (int rc = check_dims(dims)) return rc;
    const cl_tuning& tun = tuning_of(dims);
    name_reset(tun);
    if (int rc = check_ptr(params, "params")) return rc;
    if (int rc = check_ptr(ts, "ts")) return rc;
    if (int rc = check_ptr(state, "state")) return rc;
    if (int rc = check_ptr(actions, "actions", dims->n_act_cols > 0)) return rc;
    if (int rc = check_ptr(out_bldg, "out_bldg")) return rc;
    if (int rc = check_ptr(out_env, "out_env")) return rc;
    if (dims->flags & CLD_KPI) {
        // the per-building accumulators read the detail planes -- except for districts of up to 32 buildings without flexible loads, whose
        // step kernel updates them itself (cl_step_lean_kpi_kernel / cl_step_full_kpi_kernel; the launch selection below refuses the rest)
        if (!(dims->flags & CLD_WRITE_DETAIL) && (dims->n_bldg > 32 || flex))
            return fail(CL_EINVAL, "CLD_KPI requires CLD_WRITE_DETAIL (except for districts of up to 32 buildings without flexible loads)");
        if (int rc = check_ptr(kpi_bldg, "kpi_bldg")) return rc;
        if (int rc = check_ptr(kpi_env, "kpi_env")) return rc;
    }
    if (t < 0 || t >= dims->n_steps) return fail(CL_ERANGE, "t=%d outside [0, %d)", t, dims->n_steps);
    if (act_stride_env == 1 && (act_stride_col % 4) != 0)
        return fail(CL_EALIGN, "act_stride_col=%lld must be a multiple of 4 floats for the coalesced layout",
                    (long long)act_stride_col);

    StepArgs a;
    a.params = params; a.ts = ts; a.state = state; a.actions = actions; a.out_bldg = out_bldg; a.out_env = out_env;
    a.kpi_bldg = kpi_bldg; a.kpi_env = kpi_env;
    a.act_stride_col = act_stride_col; a.act_stride_env = act_stride_env;
    a.n_env = dims->n_env; a.n_bldg = dims->n_bldg; a.n_steps = dims->n_steps;
    a.flags = dims->flags; a.t = t; a.env_row0 = dims->env_row0; a.env_offset = (unsigned)dims->env_offset;
    a.ld = pitch_of(dims);
    // a pitch is implemented where 2^20-env batches exist: battery + PV districts stepped by the lean / env-major / general lean kernels
    if (a.ld != a.n_env && (!(dims->flags & CLD_LEAN) || (dims->flags & (CLD_WRITE_DETAIL | CLD_KPI | CLD_F64_MAPS)) || flex || dims->n_bldg > 32))
        return fail(CL_EINVAL, "env_pitch=%d != n_env=%d is implemented for CLD_LEAN districts of up to 32 buildings without detail planes, streaming KPIs, "
                               "flexible loads or CLD_F64_MAPS", a.ld, a.n_env);
    a.flex_out = nullptr; a.n_flex_bldg = 0; a.ev_penalty_coef = 0.0f;
    a.fused_finish = 0;                       // set where the kernel that is launched can fold the chunk sums itself (district_reduce<.., FOLD>)
    // non-temporal plane stores while the launch's footprint (~40 - 60 B per (env, building) unit) stays inside the Infinity Cache
    // ... and again once it is several times that cache (17 x 1 048 576: 125 -> 115 us, 17 x 1 572 864: 196 -> 170 us): nothing of a step
    // survives in the cache until the next one anyway, and the hint keeps the stores from displacing what the step still reads.  In
    // between (footprint of the order of the cache: 17 x 262 144 +10 %, 17 x 524 288 +-4 %) plain stores win.
    const long long nt_units = (long long)dims->n_env * dims->n_bldg;
    a.nt = tun.nt_stores == 1 || (tun.nt_stores == 0 && (nt_units <= CL_NT_MAX_UNITS || nt_units >= CL_NT_STREAM_UNITS));
    const int rkind_host = (dims->flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT;
    if (rkind_host == CLR_EV && !flex) return fail(CL_EINVAL, "reward kind CLR_EV needs the flexible-load tables (cl_step_flex_f32)");
    if (flex) {
        if (int rc = check_flex(dims, flex)) return rc;
        FlexArgs fa;
        fa.f = *flex; fa.actions = actions; fa.act_stride_col = act_stride_col; fa.act_stride_env = act_stride_env;
        fa.env_row0 = dims->env_row0; fa.n_env = dims->n_env; fa.n_steps = dims->n_steps; fa.t = t; fa.env_offset = (unsigned)dims->env_offset;
        fa.want_reward = rkind_host == CLR_EV;
        fa.want_chargers = (dims->flags & CLD_WRITE_DETAIL) != 0;
        const int units = flex->n_flex_bldg + flex->n_ev;
        const unsigned gy = (unsigned)((units + 3) / 4);
        // four envs per lane once there are enough envs to fill the chip that way (and plane rows stay 16-byte aligned)
        int fvec = dims->n_env >= 16384 ? 4 : 1;
        if (tun.flex_vec) fvec = tun.flex_vec;
        const dim3 fgrid((unsigned)((dims->n_env + 64 * fvec - 1) / (64 * fvec)), gy);
#define CL_FLEX_LAUNCH(V) do { \
            name_add(tun, "cl_flex_kernel<" #V ", %s>", a.nt ? "true" : "false"); \
            if (a.nt) hipLaunchKernelGGL((cl_flex_kernel<V, true>), fgrid, dim3(256), 0, (hipStream_t)stream, fa); \
            else hipLaunchKernelGGL((cl_flex_kernel<V, false>), fgrid, dim3(256), 0, (hipStream_t)stream, fa); } while (0)
        switch (fvec) {
        case 4: CL_FLEX_LAUNCH(4); break;
        case 2: CL_FLEX_LAUNCH(2); break;
        default: CL_FLEX_LAUNCH(1); break;
        }
#undef CL_FLEX_LAUNCH
        a.flex_out = flex->flex_out; a.n_flex_bldg = flex->n_flex_bldg;
        a.ev_penalty_coef = flex->cons_params ? flex->weights[CLEW_PENALTY_COEFFICIENT] : 0.0f;
    }
    const bool full = !(dims->flags & CLD_LEAN) || (dims->flags & CLD_WRITE_DETAIL) || (tun.lean_variant & 4);   // 4: lean districts through cl_step_full_kernel (experiments)
    a.nw = tun.nw ? tun.nw : pick_nw(dims->n_bldg, 1);
    // general kernel: two buildings per wave measured fastest for the 6..16-building thermal schemas (fewer, longer waves)
    if (!tun.nw && full && dims->n_bldg >= 6 && dims->n_bldg <= 16) a.nw = (dims->n_bldg + 1) / 2;
    const bool will_chunk = dims->n_bldg > 32 && !tun.no_chunks;
    int vec = full ? 1 : pick_vec(dims->n_env, dims->n_bldg, act_stride_env == 1);
    if (will_chunk && act_stride_env == 1) {               // few envs, many buildings: width from the unit count
        const long long units = (long long)dims->n_env * dims->n_bldg;
        // (under the float64 chain the four-env pack pays off one octave later -- scripts/r06_cliffs.py, profiles/r06c_cliffs_chain.jsonl: 33 x 16 384 and
        //  128 x 4 096, both 2^19 units, 13.8 / 12.6 us at four envs per lane against 12.2 / 9.8 us at one)
        const long long lean4 = (dims->flags & CLD_F64_CHAIN) ? (1ll << 20) : (1ll << 19);
        vec = full ? (units >= (1ll << 19) && dims->n_env >= 256 ? 2 : 1) : (units >= lean4 && dims->n_env >= 512 ? 4 : 1);
        // (chain, districts just beyond the 32-building limit of the one-row kernels on batches of >= 2048 one-env tiles: two rows of ~17 buildings at four
        //  envs per lane leave every wave one or two buildings behind a long load chain -- 33 x 262 144: 117.5 us against 74 us for the UNCHUNKED general
        //  kernel at one env per lane, which the grid rule below selects by itself once the tile count reaches 2048; profiles/r06d_cliffs_chain.jsonl)
        if (!full && (dims->flags & CLD_F64_CHAIN) && dims->n_bldg <= 40 && dims->n_env >= 131072) vec = 1;
    }
    if (tun.vec) vec = tun.vec;
    // CLD_F64_MAPS: the battery map in float64 -- general and lean step kernels at one or two envs per lane (a double is two VGPRs)
    const bool f64 = dims->flags & CLD_F64_MAPS;
    // CLD_F64_CHAIN: the soc chain in float64 on the default three state planes (cl_unit.h battery_charge_chain) -- lean, env-major, general and
    // thermal-specialised step kernels
    const bool chain = dims->flags & CLD_F64_CHAIN;
    if (chain) {
        if (f64) return fail(CL_EINVAL, "CLD_F64_CHAIN and CLD_F64_MAPS are two precision models of the same map: pick one");
        if (flex) return fail(CL_EINVAL, "CLD_F64_CHAIN is not implemented for districts with flexible loads (the EV batteries of cl_flex_kernel are fp32)");
        // (streaming KPIs without the detail planes: the lean step launch updates them itself under the chain too; a thermal district needs the planes)
        if ((dims->flags & CLD_KPI) && !(dims->flags & CLD_WRITE_DETAIL) && (full || dims->n_bldg > 32))
            return fail(CL_EINVAL, "CLD_F64_CHAIN with CLD_KPI needs CLD_WRITE_DETAIL (except for battery + PV districts of up to 32 buildings)");
        if (full) vec = 1;                     // (the thermal unit around the float64 chain spills at two envs per lane)
    }
    if (f64) {
        if (flex) return fail(CL_EINVAL, "CLD_F64_MAPS is not implemented for districts with flexible loads (the EV batteries of cl_flex_kernel are fp32)");
        if ((dims->flags & CLD_KPI) && !(dims->flags & CLD_WRITE_DETAIL)) return fail(CL_EINVAL, "CLD_F64_MAPS with CLD_KPI needs CLD_WRITE_DETAIL");
        vec = full ? 1 : (vec > 2 ? 2 : vec);          // (the thermal unit with a float64 battery spills at two envs per lane)
    }
    if (flex && vec > 2 && !(!full && dims->n_bldg <= 2 * a.nw && !will_chunk && (dims->n_env + 64 * vec - 1) / (64 * vec) <= 256 && !(tun.lean_variant & 1)))
        vec = 2;                             // general-kernel FLEX instantiations exist for 1 and 2 envs per lane
    const int tile = 64 * vec;
    const unsigned grid_x = (unsigned)((dims->n_env + tile - 1) / tile);
    // Large districts (e.g. 1024 buildings x 1024 envs per GPU): a 1-D grid over env tiles would leave most CUs idle, so
    // the buildings are cut into chunks along gridDim.y and the district sums are finished by a second tiny kernel.
    a.b_chunk = dims->n_bldg; a.n_chunks = 1;
    if (dims->n_bldg > 32 && grid_x < 2048 && !tun.no_chunks) {
        long long r = ((long long)dims->n_bldg * grid_x) / (16ll * 2048);
        if (r < 1) r = 1;
        // about one 16-wave workgroup per CU when the launch is small: the 1024 x 1024 thermal shard at two envs per lane then gives
        // every wave two buildings (32 chunks x 8 env tiles = 256 workgroups, all resident at once): 16.0 vs 18.0 us with one
        // building per wave in two generations (scripts/c4_sweep.py, profiles/r02_c4_chunk_sweep.log)
        const long long per_cu = ((long long)dims->n_bldg * grid_x + 2048) / 4096;
        if (per_cu >= 2 && r < 2) r = 2;
        // (round 5) the thermal kernel with staged parameter blocks (cl_full.h LP) on batches of several workgroup generations: FEWER, LARGER
        // chunks -- one or two workgroups per CU, each wave walking 8 - 16 buildings -- instead of eight generations of two buildings per
        // wave.  1024 buildings (scripts/gpurun/r05_call12.sh, profiles/r05m_*): x 8192 envs 103.0 / 95.3 / 91.4 / 91.6 us with chunks of
        // 32 / 64 / 128 / 256; x 4096: 55.0 / 50.9 / 48.7 / 78.4 (128 workgroups leave half the CUs idle); x 2048: 26.6 / 30.1 / 36.4 / 61.0
        // and x 1024: 13.2 / 21.5 / 34.5 / 60.2 -- those stay at 32.  (Battery + PV districts, cl_step_kernel: 32 stays best, 8192 envs:
        // 63.8 / 60.3 / 64.9 / 67.4 us with 24 / 32 / 48 / 64.)
        const bool lp_shape = !(dims->flags & CLD_LEAN) && !(tun.lean_variant & 4) && !flex && !(dims->flags & (CLD_WRITE_DETAIL | CLD_F64_MAPS)) && vec == 2 &&
                              tun.full_variant != 1 && tun.full_variant != 3;
        if (lp_shape && dims->n_bldg >= 256 && (long long)dims->n_bldg * grid_x >= 32ll * 1024) {     // (measured on 1024 buildings; smaller districts keep the rule above)
            long long want = ((long long)dims->n_bldg * grid_x + 255) / 256;          // buildings per chunk for 256 workgroups ...
            r = 2; while (16 * r < want && r < 8) r *= 2;                            // ... as a power of two, 128 at most (40 KB of staged blocks)
        }
        // (round 6) the thermal kernel around the float64 chain (one env per lane, parameter blocks through the constant cache): ONE workgroup per CU --
        // 256 workgroups, chunks of up to 256 buildings.  1024 buildings x 1024 / 2048 / 4096 / 8192 envs with chunks of 32 / 64 / 128 / 256:
        // 20.2 / 17.7 / 27.4 / 47.6, 37.4 / 33.5 / 31.3 / 48.8, 71.3 / 68.7 / 67.5 / 65.8, 149.7 / 137.3 / 129.5 / 125.4 us
        // (scripts/gpurun/r06_call15.sh, profiles/r06o_*).
        const bool chain_full_shape = chain && !(dims->flags & CLD_LEAN) && !flex && !(dims->flags & (CLD_WRITE_DETAIL | CLD_F64_MAPS)) && tun.full_variant != 1;
        // 128 .. 1024 buildings x 1024 .. 65 536 envs (scripts/r06_chunk_sweep.py, profiles/r06p_chunks_*.jsonl): the rule is within 3 % of the best chunk size
        // of every cell -- 128 x 16 384 36.4 -> 25.6 us, 256 x 16 384 66.9 -> 49.2, 512 x 16 384 153 -> 126, 1024 x 16 384 317 -> 249 (chunks as large as
        // the district = one workgroup row, no second launch)
        if (chain_full_shape && dims->n_bldg >= 128 && (long long)dims->n_bldg * grid_x >= 16ll * 1024) {
            const long long want = ((long long)dims->n_bldg * grid_x + 255) / 256;
            r = 2; while (16 * r < want && r < 16) r *= 2;
        }
        // (fp32 map, same sweep: up to 256 buildings x 65 536 envs in ONE workgroup row -- 128 buildings 113 -> 98 us, 256 buildings 204 -> 182 us; at
        //  16 384 envs the chunked launch stays ahead, 29.0 vs 32.4 and 49.0 vs 59.1 us)
        if (lp_shape && dims->n_bldg <= 256 && grid_x >= 512) r = 16;
        // ... whose plane stores take the non-temporal hint at every batch size (the footprint rule above is the battery + PV kernels': 1024 x
        // 8192 envs 101.2 -> 100.0 us, x 4096 54.3 -> 53.3 us, chunks of 128: 91.4 -> 89.1 us; profiles/r05l_*, r05m_*)
        if (lp_shape && tun.nt_stores == 0) a.nt = 1;
        a.b_chunk = tun.b_chunk > 0 ? tun.b_chunk : (int)(16 * r);
        a.n_chunks = (dims->n_bldg + a.b_chunk - 1) / a.b_chunk;
        if (tun.b_chunk <= 0 && a.n_chunks > 1) {
            // balanced chunks (round 6): 33 buildings were cut 16 + 16 + 1 -- a third workgroup row per env tile for one building; now round(33 / 16) = 2
            // rows of 17 (one wave of the sixteen walks two buildings).  Districts that divide evenly (1024 / 32) keep their geometry.
            const int nc = (int)((2ll * dims->n_bldg + a.b_chunk) / (2ll * a.b_chunk));       // round(n_bldg / b_chunk)
            if (nc >= 2) { a.n_chunks = nc; a.b_chunk = (dims->n_bldg + nc - 1) / nc; a.n_chunks = (dims->n_bldg + a.b_chunk - 1) / a.b_chunk; }
        }
        if (a.n_chunks == 1) a.b_chunk = dims->n_bldg;
        else a.nw = (tun.b_chunk > 0 && tun.nw > 0) ? tun.nw : 16;
        // the reserved plane holds the chunk partial sums (twice under the deferred finish), the tickets of the in-launch fold and, in its
        // last 16 bytes, the marker words EVERY chunked launch touches (a non-deferring one clears its step's marker)
        // (the second buffer only where the launch can defer at all: finish = 3 on a launch that keeps the second cl_finish launch needs one)
        const long long scratch_words = (long long)a.n_chunks * NQ * dims->n_env + (dims->n_env + 63) / 64 + 4;
        if (scratch_words > (long long)dims->n_bldg * dims->n_env)
            return fail(CL_EINVAL, "b_chunk=%d leaves no room for the %d chunk partial sums, their tickets and the marker words", a.b_chunk, a.n_chunks);
    }
    if (a.n_chunks > 1 && rkind_host == CLR_EV)
        return fail(CL_EINVAL, "reward kind CLR_EV is not implemented for building-chunked launches (n_bldg=%d)", dims->n_bldg);
    // Deferred finish (cl_tuning.finish = 3): the launch folds the PREVIOUS step's chunk sums and leaves its own for the next launch or for
    // cl_finish_f32 (district_reduce).  Only where nothing of the path reads out_env inside the step: no coupled reward (MARL's per-building
    // rewards need the district net of the same step, reward_function.py:132-143; the EV reward likewise), no streaming KPIs, no flexible
    // loads, and the kernels that carry the fold (the FOLD instantiations below); a 16-wave workgroup folds at most 64 district sums of at most 64 chunks, 1024 partial sums in all,
    // and the reserved plane has to hold both buffers and the marker words.  Anything else keeps the second launch.
    const int fold_per_row = a.n_chunks > 1 ? (NQ * tile + a.n_chunks - 1) / a.n_chunks : 0;
    // (battery + PV districts keep the 16-sum limit: where more sums per row would be needed -- 1024 x 4096 / 8192 envs at four envs per lane --
    //  the folding instantiation's 107 registers cost more than the second launch: 32.8 vs 32.0 us, 65.2 vs 61.3 us, profiles/r05n_*)
    const int fold_w = fold_per_row <= 16 ? 16 : fold_per_row <= 32 ? 32 : 64;          // row width of the exchange tile (fold_shift)
    const bool can_defer = a.n_chunks > 1 && tun.finish == 3 && rkind_host != CLR_MARL && rkind_host != CLR_EV && !flex &&
                           !(dims->flags & (CLD_KPI | CLD_F64_MAPS | CLD_WRITE_DETAIL)) && (!chain || !full) &&      // (the float64 chain: the battery + PV chunk kernel carries the fold; the thermal chain kernel does not)
                           fold_per_row <= (full ? 64 : 16) && a.n_chunks * fold_w <= 1024 &&
                           a.nw == 16 && a.n_chunks <= 64 &&
                           2ll * a.n_chunks * NQ * dims->n_env + (dims->n_env + 63) / 64 + 4 <= (long long)dims->n_bldg * dims->n_env;
    const dim3 grid(grid_x, a.n_chunks);
    const bool det = dims->flags & CLD_WRITE_DETAIL;
    // streaming KPIs of thermal / outage districts (and of any district stepped with detail planes) inside the step launch:
    // cl_step_full_kpi_kernel (cl_full.h); cl_tuning.kpi_passes = 1 keeps the separate cl_kpi_kernel pass (A/B), 2 the two round-1 passes
    // (up to 128 buildings: their baselines of one env tile sit in LDS, 256 B per building)
    const bool kpi_full = (dims->flags & CLD_KPI) && full && !flex && !f64 && !chain && a.n_chunks == 1 && vec == 1 && tun.full_variant != 1 && tun.kpi_passes == 0 &&
                          dims->n_bldg <= 128;
    // ... whose waves should all be resident at once (16 per CU at its 119 registers): as many waves per workgroup as that allows, at least
    // two (9 x 65 536: four waves 18.8 us, the step-only default of five -- two generations -- 23.6 us; profiles/r03_kpi_in_step_probe.log)
    if (kpi_full && !tun.nw) {
        const long long fit = (16ll * 256) / grid_x;
        a.nw = (int)(fit < 2 ? 2 : fit > 16 ? 16 : fit);
        if (a.nw > dims->n_bldg) a.nw = dims->n_bldg;
    }
    // thermal kernel with the parameter blocks of the workgroup's buildings staged in LDS
    // (for the building-chunked launches only -- a workgroup of the 9 x 65 536 launch would wait for the staging round trip before it
    //  can issue its plane loads, while its scalar reads hit the constant cache: 10.7 vs 8.7 us; full_variant = 2 forces it, 3 forbids it)
    // (not under the float64 chain unless forced: its one-env-per-lane thermal kernel reads the blocks through the constant cache faster -- chunked 1024-,
    //  512-, 256-building districts 1.06 - 1.21 x, profiles/r06c_cliffs_chain.jsonl -- and a 512-building chunk's 160 KB of staged blocks do not exist)
    const bool lp = full && !flex && !det && !f64 && tun.full_variant != 1 && tun.full_variant != 3 && vec <= 2 &&
                    ((a.n_chunks > 1 && !chain) || tun.full_variant == 2) && (size_t)a.b_chunk * CL_LP_WORDS * sizeof(uint32_t) <= 96 * 1024;
    const size_t lds = (size_t)a.nw * NQ * tile * sizeof(float) + (lp ? (size_t)a.b_chunk * CL_LP_WORDS * sizeof(uint32_t) : 0) +
                       (can_defer ? 1024 * sizeof(float) : 0);          // (+ the [chunks][16 / 32 / 64 sums] exchange tile of the deferred fold)
    const dim3 block(64 * a.nw);
    hipStream_t s = (hipStream_t)stream;
    // Thermal districts whose batch can be cut into ONE 16-wave workgroup per CU: a workgroup takes `tiles` 128-env tiles (two envs per
    // lane) and deals its tiles x B (tile, building) items to the 16 waves in order (cl_step_full_tp_kernel) -- the items divide over
    // the four SIMDs where the B buildings of one tile do not, and the whole launch is resident at once.  scripts/tp_sweep.py,
    // scripts/tp_sweep2.py (profiles/r02_tp_sweep*.log), one-tile kernel -> this one: 9 x 65 536 8.5 -> 7.7 us, 9 x 131 072 17.3 -> 14.3,
    // 9 x 262 144 29.4 -> 28.0, 12 x 65 536 11.8 -> 9.2, 16 x 65 536 13.4 -> 11.7, 6 x 65 536 7.0 -> 6.6, 3 x 262 144 11.9 -> 10.7; with
    // fewer than ~12 items per workgroup (3 x 65 536) or with more / fewer workgroups than CUs the one-tile kernel wins and stays.
    // full_variant: 5 forces it (tun.vec = envs per lane, tun.nw = waves, tun.b_chunk = tiles), 3 forbids it.
    const bool tp_forced = tun.full_variant == 5;
    // small batches (193 .. 256 one-env-per-lane tiles, i.e. up to 16 384 envs): one tile per workgroup, one WAVE per building --
    // 9 x 16 384 5.62 -> 5.01 us, 6 x 16 384 5.50 -> 4.41, 12 x 16 384 6.15 -> 5.25, 16 x 16 384 6.26 -> 5.84 (scripts/tp_small_probe.py)
    const unsigned tiles1 = (unsigned)((dims->n_env + 63) / 64);
    const bool tp_small = !tp_forced && tiles1 > 192 && tiles1 <= 256 && dims->n_bldg >= 6 && dims->n_bldg <= 16;
    const int tp_vec = (tp_forced && tun.vec == 1) || tp_small || chain ? 1 : 2;       // (the float64 chain spills at two envs per lane)
    const int tp_auto_tiles = tp_small ? 1 : (int)((dims->n_env + 256 * 64 * tp_vec - 1) / (256 * 64 * tp_vec));      // one workgroup per CU
    // (chain, round 6: where one workgroup per CU would need more tiles than LDS holds -- 17 / 20 thermal buildings x 262 144 envs -- four tiles per
    //  workgroup in several generations still beat the one-tile kernel 1.36 x / 1.16 x: profiles/r06c_cliffs_chain.jsonl)
    const size_t tp_tile_bytes = ((size_t)dims->n_bldg * NQ + 1) * 64 * tp_vec * sizeof(float);
    const bool tp_capped = chain && !tp_forced && !tp_small && (size_t)tp_auto_tiles * tp_tile_bytes > 150 * 1024 && 4 * tp_tile_bytes <= 150 * 1024 && dims->n_bldg >= 6;
    const int tp_tiles = tp_forced ? (tun.b_chunk > 0 ? tun.b_chunk : CL_ROW0_BLOCK / (64 * tp_vec)) : tp_capped ? 4 : tp_auto_tiles;
    const int tp_nw = tp_forced && tun.nw ? tun.nw : (tp_small ? dims->n_bldg : 16);
    const unsigned tp_grid = (unsigned)((dims->n_env + tp_tiles * 64 * tp_vec - 1) / (tp_tiles * 64 * tp_vec));
    const size_t tp_lds = ((size_t)tp_tiles * dims->n_bldg * NQ + tp_tiles) * 64 * tp_vec * sizeof(float);
    bool tp_kernel = full && !flex && !(dims->flags & CLD_WRITE_DETAIL) && !kpi_full && a.n_chunks == 1 && dims->n_bldg <= 32 && tp_lds <= 150 * 1024 &&
                     (!dims->env_row0 || CL_ROW0_BLOCK % (tp_tiles * 64 * tp_vec) == 0);    // one episode offset per workgroup: no workgroup straddles two blocks
    if (tp_forced) {
        if (!tp_kernel || tp_nw > 16)
            return fail(CL_EINVAL, "full_variant = 5: %d tiles x %d envs per lane x %d waves is not a launch of cl_step_full_tp_kernel for this district", tp_tiles, tp_vec, tp_nw);
    } else tp_kernel = tp_kernel && tun.full_variant == 0 && !tun.vec && !tun.nw && (tp_small || tp_tiles * dims->n_bldg >= 12) && ((tp_grid > 192 && tp_grid <= 256) || tp_capped);
    // (up to 480 workgroups -- two 9-wave workgroups per CU are resident at once, so up to 512 the launch is still ONE generation: re-measured
    //  at the end of round 5, after the latency-ordered kernel lost the non-temporal hint on its loads (scripts/gpurun/r05_call24.sh,
    //  profiles/r05_nt_loads/r05y.log), 17 buildings x 98 304 / 106 496 / 114 688 / 131 072 / 163 840 / 196 608 / 262 144 envs: 10.6 / 11.0 /
    //  11.8 / 15.5 / 19.2 / 22.6 / 31.7 us against 13.2 / 13.2 / 13.5 / 15.6 / 19.2 / 21.3 / 26.2 us for the env-major kernel -- the old rule
    //  (352 workgroups, env-major from 106 496 envs) had the general kernel at 98 304 envs, 13.0 us)
    // streaming KPIs without the detail planes: the lean kernel updates the per-building accumulators itself, at any grid size
    const bool kpi_lean = (dims->flags & CLD_KPI) && !(dims->flags & CLD_WRITE_DETAIL) && !kpi_full;
    // cl_step_observe_f32 on a thermal / outage district: the step launch fills the compact observation itself (cl_full.h OBS) where it is one
    // workgroup row of the one-env-per-lane kernel or the multi-tile kernel, without detail planes / KPIs / a coupled reward
    const bool obs_full = of_full && full && !flex && !det && !f64 && !(dims->flags & CLD_KPI) && a.n_chunks == 1 && rkind_host != CLR_MARL && tuning_of(dims).obs_variant == 0;
    // Env-major or building-major above one wave generation?  Re-measured in round 6, both kernels alternating in ONE process (scripts/r06_lean_vs_envmajor.py,
    // profiles/r06_lean_vs_envmajor*.log; a process lands in a +- 4 % band, which is what hid this in round 5):
    //  * the env-major kernel wins where the step's footprint is of the order of the Infinity Cache -- 17 x 262 144: 25.8 / 30.7 us (fp32 / chain) against 27.9 /
    //    32.2 for the latency-ordered building-major kernel at four envs per lane; 9 x 262 144: 16.1 / 18.8 against 19.2 / 21.4;
    //  * far beyond the cache the building-major kernel's 16-byte accesses win again: 17 x 1 048 576 -- the HBM-true shape of the bench line -- 104.6 - 111.8 us
    //    against 113.5 - 127.6 us (fp32) and 113.5 - 120.4 against 125.6 - 134.4 (chain) in three processes, 17 x 2 097 152 222 - 224 / 236 against 225 - 247 /
    //    245 - 261, 20 x 1 048 576 (fp32) 126 - 129 against 133 - 140;
    //  * under the float64 chain the building-major kernel holds on longer below: 17 x 147 456 / 163 840 / 180 224 20.5 / 21.8 / 22.8 us against 23.4 / 24.2 /
    //    25.2 (196 608: 24.4 - 25.6 against 25.8 - 26.5; 229 376: even); 9 and 6 buildings: 131 072 envs 10.1 / 7.6 against 11.1 / 8.6, even or behind from 147 456.
    //  * (scripts/r06_stream_map.py, 6 .. 20 buildings x 393 216 .. 2 097 152 envs, five variants side by side, medians of three rounds; profiles/r06_stream_map*.jsonl)
    //    from 12 buildings and 8 Mi units the building-major kernel WITH non-temporal stores is the best or within 3 % of it in 25 of 30 cells -- 12 x 786 432 /
    //    1 048 576: 57.0 / 75.0 us against 62.8 / 84.3 (fp32), 57.0 / 75.0 against 68.9 / 94.0 (chain); 20 x 524 288 / 786 432 (fp32): 61.2 / 93.2 against 72.8 / 106.9;
    //    6 and 9 buildings are mixed and keep the env-major kernel.
    const bool stream_lean = dims->n_bldg >= 12 && (long long)dims->n_bldg * dims->n_env >= (8ll << 20);
    const int em_min = !chain ? 122880 : dims->n_bldg >= 16 ? 196608 : 131072;
    const bool em_auto = dims->n_env > em_min && !(chain && dims->n_bldg > 17) && !stream_lean;
    const bool lean_beyond = tun.envmajor == 0 && !full && dims->n_bldg <= 20 && dims->n_env > 122880 && !em_auto;      // (what the env-major rule no longer takes)
    const bool lean_shape = a.n_chunks == 1 && dims->n_bldg <= 2 * a.nw && (grid_x <= 480 || (tun.lean_variant & 2) || kpi_lean || lean_beyond) &&
                            !((tun.lean_variant & 1) && !kpi_lean);
    if (lean_beyond && stream_lean && lean_shape && tun.nt_stores == 0) a.nt = 1;      // (8 .. 16 Mi units: the footprint rule above says plain stores -- measured on the env-major kernel)
    // without the detail planes only cl_step_lean_kpi_kernel updates the per-building accumulators (and writes the baseline plane
    // cl_kpi_env_kernel sums): a launch shape that cannot take it must not silently leave them stale
    if (kpi_lean && (full || flex || !lean_shape))
        return fail(CL_EINVAL, "CLD_KPI without CLD_WRITE_DETAIL needs a step launch that updates the accumulators itself (battery + PV: n_bldg=%d <= 2 x nw=%d "
                               "waves, no chunks; thermal: one env per lane, no chunks, no flexible loads, no CLD_F64_MAPS): drop the cl_tuning override or set CLD_WRITE_DETAIL",
                    dims->n_bldg, a.nw);
    // (chain: the 20-building instantiation holds 140 registers -- three waves per SIMD -- and loses to the building-major kernel, 41.6 vs 33.6 us at
    //  20 x 262 144: only districts of up to 17 buildings go env-major by themselves; profiles/r06c_cliffs_chain.jsonl)
    const bool envmajor_shape = !full && a.n_chunks == 1 && dims->n_bldg <= 20 && !kpi_lean &&
                                (tun.envmajor == 1 || (tun.envmajor == 0 && em_auto));
    if (dims->flags & CLD_CHECK) {
        // debug mode: the general kernel with the reference's assertions compiled in, one env per lane (include/citylearn_amd.h CLD_CHECK)
        if (!det || (dims->flags & CLD_DETAIL_MIN) || a.n_chunks > 1 || kpi_full)
            return fail(CL_EINVAL, "CLD_CHECK needs CLD_WRITE_DETAIL (all planes), a district of up to 32 buildings (the violation words use the reserved plane) "
                                   "and, with CLD_KPI, the separate KPI launch (cl_tuning.kpi_passes = 1)");
        const dim3 grid1((unsigned)((dims->n_env + 63) / 64));
        const size_t lds1 = (size_t)a.nw * NQ * 64 * sizeof(float);
        name_add(tun, "cl_step_kernel<1, true, true, %s, %d, false, true>", flex ? "true" : "false", chain ? 2 : f64 ? 1 : 0);
        if (flex) hipLaunchKernelGGL((cl_step_kernel<1, true, true, true, 0, false, true>), grid1, block, lds1, s, a);
        else if (chain) hipLaunchKernelGGL((cl_step_kernel<1, true, true, false, 2, false, true>), grid1, block, lds1, s, a);
        else if (f64) hipLaunchKernelGGL((cl_step_kernel<1, true, true, false, 1, false, true>), grid1, block, lds1, s, a);
        else hipLaunchKernelGGL((cl_step_kernel<1, true, true, false, 0, false, true>), grid1, block, lds1, s, a);
    } else if (chain) {
        if (envmajor_shape) {
            const dim3 egrid((unsigned)((dims->n_env + 255) / 256));
            const int enb = dims->n_bldg <= 17 ? 17 : 20;
            name_add(tun, "cl_step_envmajor_kernel<%d, %s, 1, 2>", enb, a.nt ? "true" : "false");
            if (enb == 17) { if (a.nt) hipLaunchKernelGGL((cl_step_envmajor_kernel<17, true, 1, 2>), egrid, dim3(256), 0, s, a);
                             else hipLaunchKernelGGL((cl_step_envmajor_kernel<17, false, 1, 2>), egrid, dim3(256), 0, s, a); }
            else { if (a.nt) hipLaunchKernelGGL((cl_step_envmajor_kernel<20, true, 1, 2>), egrid, dim3(256), 0, s, a);
                   else hipLaunchKernelGGL((cl_step_envmajor_kernel<20, false, 1, 2>), egrid, dim3(256), 0, s, a); }
        } else if (!full && lean_shape) {
            const bool obs_fused = of && rkind_host != CLR_MARL && !kpi_lean;      // (MARL's reward plane is finished after the sweep the tile is filled in)
            const size_t lds_x = lds + (kpi_lean ? CL_OBS_FUSED_BLDG * sizeof(float) : obs_fused ? (size_t)tile * of->pitch * sizeof(float) : 0);
#define CL_CHAIN_CASE(V) case V: \
            name_add(tun, "%s<" #V ", %s>", kpi_lean ? "cl_step_lean_kpi_chain_kernel" : obs_fused ? "cl_step_lean_obs_chain_kernel" : "cl_step_lean_chain_kernel", a.nt ? "true" : "false"); \
            if (kpi_lean) { \
                if (a.nt) hipLaunchKernelGGL((cl_step_lean_kpi_chain_kernel<V, true>), grid, block, lds_x, s, a); \
                else hipLaunchKernelGGL((cl_step_lean_kpi_chain_kernel<V, false>), grid, block, lds_x, s, a); \
            } else if (obs_fused) { \
                if (a.nt) hipLaunchKernelGGL((cl_step_lean_obs_chain_kernel<V, true>), grid, block, lds_x, s, a, *of); \
                else hipLaunchKernelGGL((cl_step_lean_obs_chain_kernel<V, false>), grid, block, lds_x, s, a, *of); \
                *fused = true; \
            } else if (a.nt) hipLaunchKernelGGL((cl_step_lean_chain_kernel<V, true>), grid, block, lds, s, a); \
            else hipLaunchKernelGGL((cl_step_lean_chain_kernel<V, false>), grid, block, lds, s, a); \
            break;
            switch (vec) {
            CL_CHAIN_CASE(1) CL_CHAIN_CASE(2) CL_CHAIN_CASE(4)
            default: return fail(CL_EINVAL, "bad vec %d", vec);
            }
#undef CL_CHAIN_CASE
        } else if (!full && a.n_chunks > 1 && tun.finish != 2 && (vec == 1 || vec == 2 || vec == 4) && !(tun.lean_variant & 16)) {
            // building-chunked battery + PV districts: the latency-ordered chunk kernel around the float64 chain (deferred fold where it applies)
            a.fused_finish = can_defer ? 2 : 0;
            name_add(tun, "cl_step_lean_chunk_kernel<%d, %s, %s, 2>", vec, a.nt ? "true" : "false", can_defer ? "true" : "false");
            if (int rc = launch_lean_chunk<2>(vec, a.nt, can_defer, grid, block, lds, s, a)) return rc;
        } else if (tp_kernel) {
            // thermal districts, several env tiles per workgroup (cl_step_full_tp_kernel's launch shape)
            a.nw = tp_nw;
            const dim3 g2(tp_grid), b2(64 * a.nw);
            if (obs_full && tp_lds + (size_t)tp_tiles * 64 * of_full->pitch * sizeof(float) <= 150 * 1024) {      // (a tile that does not fit: the two launches)
                const size_t l2 = tp_lds + (size_t)tp_tiles * 64 * of_full->pitch * sizeof(float);
                if (int rc = launch_tp_obs<1, 2>(a.nt, g2, b2, l2, s, a, tp_tiles, *of_full, tun)) return rc;
                *fused = true;
            } else {
            name_add(tun, "cl_step_full_tp_chain_kernel<1, 4, %s>", a.nt ? "true" : "false");
            if (a.nt) hipLaunchKernelGGL((cl_step_full_tp_chain_kernel<1, 4, true>), g2, b2, tp_lds, s, a, tp_tiles);
            else hipLaunchKernelGGL((cl_step_full_tp_chain_kernel<1, 4, false>), g2, b2, tp_lds, s, a, tp_tiles);
            }
        } else if (full && tun.full_variant != 1) {
            // thermal / outage districts: the pack-generic kernel of cl_full.h at one env per lane (parameter blocks staged in LDS where chunked)
            if (det) CL_LAUNCH_NT(cl_step_full_chain_kernel, 1, true, 1024, 4, false);
            else if (obs_full && !lp) {
                if (int rc = launch_full_obs<2>(a.nt, grid, block, lds + (size_t)64 * of_full->pitch * sizeof(float), s, a, *of_full, tun)) return rc;
                *fused = true;
            } else if (lp) { const dim3 grid_xy = grid; { const dim3 grid = CL_SWAP_GRID ? dim3(grid_xy.y, grid_xy.x) : grid_xy; CL_LAUNCH_NT(cl_step_full_chain_kernel, 1, false, 1024, 4, true); } }
            else CL_LAUNCH_NT(cl_step_full_chain_kernel, 1, false, 1024, 4, false);
        } else {
            name_add(tun, "cl_step_kernel<%d, %s, %s, false, 2, false>", vec, full ? "true" : "false", full && det ? "true" : "false");
            if (full && det) hipLaunchKernelGGL((cl_step_kernel<1, true, true, false, 2>), grid, block, lds, s, a);
            else if (full) hipLaunchKernelGGL((cl_step_kernel<1, true, false, false, 2>), grid, block, lds, s, a);
            else if (vec == 1) hipLaunchKernelGGL((cl_step_kernel<1, false, false, false, 2>), grid, block, lds, s, a);
            else if (vec == 2) hipLaunchKernelGGL((cl_step_kernel<2, false, false, false, 2>), grid, block, lds, s, a);
            else hipLaunchKernelGGL((cl_step_kernel<4, false, false, false, 2>), grid, block, lds, s, a);
        }
    } else if (f64) {
        if (!full && lean_shape) {
            if (vec == 1) CL_LAUNCH_NT(cl_step_lean_f64_kernel, 1); else CL_LAUNCH_NT(cl_step_lean_f64_kernel, 2);
        } else {
            name_add(tun, "cl_step_kernel<%d, %s, %s, false, 1, false>", vec, full ? "true" : "false", full && det ? "true" : "false");
            if (full && det) hipLaunchKernelGGL((cl_step_kernel<1, true, true, false, 1>), grid, block, lds, s, a);
            else if (full) hipLaunchKernelGGL((cl_step_kernel<1, true, false, false, 1>), grid, block, lds, s, a);
            else if (vec == 1) hipLaunchKernelGGL((cl_step_kernel<1, false, false, false, 1>), grid, block, lds, s, a);
            else hipLaunchKernelGGL((cl_step_kernel<2, false, false, false, 1>), grid, block, lds, s, a);
        }
    } else if (flex && !full && lean_shape) {
        switch (vec) {
        case 1: CL_LAUNCH_NT(cl_step_lean_kernel, 1, true); break;
        case 2: CL_LAUNCH_NT(cl_step_lean_kernel, 2, true); break;
        case 4: CL_LAUNCH_NT(cl_step_lean_kernel, 4, true); break;
        default: return fail(CL_EINVAL, "bad vec %d", vec);
        }
    } else if (flex) {
        // districts with chargers / washing machines: the FLEX instantiations of the general kernel
        if (vec > 2) return fail(CL_EINVAL, "bad vec %d for the flexible-load step", vec);
        const dim3& grid_f = grid;
        const size_t lds_f = lds;
        name_add(tun, "cl_step_kernel<%d, %s, %s, true, 0, false>", vec, full ? "true" : "false", full && det ? "true" : "false");
        if (full && det) {
            if (vec == 1) hipLaunchKernelGGL((cl_step_kernel<1, true, true, true>), grid_f, block, lds_f, s, a);
            else hipLaunchKernelGGL((cl_step_kernel<2, true, true, true>), grid_f, block, lds_f, s, a);
        } else if (full) {
            if (vec == 1) hipLaunchKernelGGL((cl_step_kernel<1, true, false, true>), grid_f, block, lds_f, s, a);
            else hipLaunchKernelGGL((cl_step_kernel<2, true, false, true>), grid_f, block, lds_f, s, a);
        } else {
            if (vec == 1) hipLaunchKernelGGL((cl_step_kernel<1, false, false, true>), grid_f, block, lds_f, s, a);
            else hipLaunchKernelGGL((cl_step_kernel<2, false, false, true>), grid_f, block, lds_f, s, a);
        }
    } else if (tp_kernel) {
        // several env tiles per workgroup (cl_step_full_tp_kernel, cl_full.h); forced: tun.vec = envs per lane, tun.nw = waves, tun.b_chunk = tiles
        a.nw = tp_nw;
        const dim3 g2(tp_grid), b2(64 * a.nw);
        const size_t l2 = tp_lds;
        if (obs_full && l2 + (size_t)tp_tiles * 64 * tp_vec * of_full->pitch * sizeof(float) <= 150 * 1024) {
            const size_t lo = l2 + (size_t)tp_tiles * 64 * tp_vec * of_full->pitch * sizeof(float);
            if (int rc = tp_vec == 2 ? launch_tp_obs<2, 0>(a.nt, g2, b2, lo, s, a, tp_tiles, *of_full, tun) : launch_tp_obs<1, 0>(a.nt, g2, b2, lo, s, a, tp_tiles, *of_full, tun)) return rc;
            *fused = true;
        } else {
        name_add(tun, "cl_step_full_tp_kernel<%d, 4, %s>", tp_vec, a.nt ? "true" : "false");
        if (tp_vec == 2) {
            if (a.nt) hipLaunchKernelGGL((cl_step_full_tp_kernel<2, 4, true>), g2, b2, l2, s, a, tp_tiles);
            else hipLaunchKernelGGL((cl_step_full_tp_kernel<2, 4, false>), g2, b2, l2, s, a, tp_tiles);
        } else {
            if (a.nt) hipLaunchKernelGGL((cl_step_full_tp_kernel<1, 4, true>), g2, b2, l2, s, a, tp_tiles);
            else hipLaunchKernelGGL((cl_step_full_tp_kernel<1, 4, false>), g2, b2, l2, s, a, tp_tiles);
        }
        }
    } else if (full && tun.full_variant != 1 && vec <= 2) {
        // thermal / outage districts: the pack-generic kernel of cl_full.h
        const bool small = block.x <= 576;
        if (kpi_full) {
            const size_t lds_k = lds + (size_t)dims->n_bldg * tile * sizeof(float);      // + the per-building baselines of the tile
            name_add(tun, "cl_step_full_kpi_kernel<%s>", a.nt ? "true" : "false");
            if (a.nt) hipLaunchKernelGGL((cl_step_full_kpi_kernel<true>), grid, block, lds_k, s, a);
            else hipLaunchKernelGGL((cl_step_full_kpi_kernel<false>), grid, block, lds_k, s, a);
        } else if (det) {
            if (vec == 1) CL_LAUNCH_NT(cl_step_full_kernel, 1, true, 1024, 4, false);
            else if (small) CL_LAUNCH_NT(cl_step_full_kernel, 2, true, 576, 3, false);
            else CL_LAUNCH_NT(cl_step_full_kernel, 2, true, 1024, 4, false);
        } else if (lp) {
            // parameter blocks staged in LDS (cl_full.h); full_variant = 3 keeps them in SGPRs (tests, A/B)
            a.fused_finish = (a.n_chunks > 1 && vec == 2 && !small) ? (tun.finish == 2 ? 1 : can_defer ? 2 : 0) : 0;
            const dim3 grid_xy = grid;
            {
                const dim3 grid = CL_SWAP_GRID ? dim3(grid_xy.y, grid_xy.x) : grid_xy;       // the LP instantiations read (building chunk, env tile) from blockIdx: district_reduce's SWAP note
                if (vec != 1 && !small && lds > 64 * 1024) {
                    // (chunks of 128 buildings: 40 KB of staged blocks + 32 KB of reduction rows + the exchange tile -- more dynamic LDS than a
                    //  kernel gets without opting in where the runtime enforces the 64 KB default)
                    const void* fn = a.nt ? reinterpret_cast<const void*>(cl_step_full_kernel<2, false, 1024, 4, true, true>)
                                          : reinterpret_cast<const void*>(cl_step_full_kernel<2, false, 1024, 4, true, false>);
                    if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess)
                        return hip_fail(e, "hipFuncSetAttribute(cl_step_full_kernel<2, .., LP>)");
                }
                if (vec == 1) CL_LAUNCH_NT(cl_step_full_kernel, 1, false, 1024, 5, true);
                else if (!small) CL_LAUNCH_NT(cl_step_full_kernel, 2, false, 1024, 4, true);
            }
            if (vec != 1 && small) CL_LAUNCH_NT(cl_step_full_kernel, 2, false, 576, 5, false);      // (96 VGPRs do not hold the staged operands: 61 scratch accesses)
        } else {
            if (vec == 1 && obs_full) {
                if (int rc = launch_full_obs<0>(a.nt, grid, block, lds + (size_t)64 * of_full->pitch * sizeof(float), s, a, *of_full, tun)) return rc;
                *fused = true;
            } else
#ifdef CL_TRACE                  // the stamps need a few registers: five waves per SIMD (what the 9-building launch holds) instead of six
            if (vec == 1) CL_LAUNCH_NT(cl_step_full_kernel, 1, false, 1024, 5, false);
#else
            if (vec == 1) CL_LAUNCH_NT(cl_step_full_kernel, 1, false, 1024, 6, false);
#endif
            else if (small) CL_LAUNCH_NT(cl_step_full_kernel, 2, false, 576, 5, false);
            else CL_LAUNCH_NT(cl_step_full_kernel, 2, false, 1024, 5, false);
        }
    } else if (full && det) {
        name_add(tun, "cl_step_kernel<%d, true, true, false, 0, false>", vec);
        switch (vec) {
        case 1: hipLaunchKernelGGL((cl_step_kernel<1, true, true>), grid, block, lds, s, a); break;
        case 2: hipLaunchKernelGGL((cl_step_kernel<2, true, true>), grid, block, lds, s, a); break;
        default: return fail(CL_EINVAL, "bad vec %d", vec);
        }
    } else if (full) {
        name_add(tun, "cl_step_kernel<%d, true, false, false, 0, false>", vec);
        switch (vec) {
        case 1: hipLaunchKernelGGL((cl_step_kernel<1, true, false>), grid, block, lds, s, a); break;
        case 2: hipLaunchKernelGGL((cl_step_kernel<2, true, false>), grid, block, lds, s, a); break;
        // (four envs per lane is not instantiated for the thermal unit: 92 bytes of scratch per lane, never selected by the library)
        default: return fail(CL_EINVAL, "bad vec %d", vec);
        }
    } else if (envmajor_shape) {
        // two or more waves per SIMD: the env-major kernel (bench.py --envs-per-gpu: 17 x 131 072 17.0 vs 18.5 us,
        // 17 x 262 144 28.2 vs 32.5 us, 17 x 1 048 576 136 vs 157 us; at 17 x 65 536 -- one wave per SIMD, nothing to hide the
        // per-building dependency chain behind -- 13.1 vs 8.0 us)
        const dim3 egrid((unsigned)((dims->n_env + 255) / 256));
        // envs per lane (cl_tuning.vec: 1 or 2) and the compile-time bound on the buildings held in flight (17 = the 2022 challenge's
        // district: three fewer register quadruples than the general 20)
        const int evec = tun.vec == 2 && act_stride_env == 1 ? 2 : 1;
        const int enb = dims->n_bldg <= 17 && tun.lean_variant != 8 ? 17 : 20;
        name_add(tun, "cl_step_envmajor_kernel<%d, %s, %d, 0>", enb, a.nt ? "true" : "false", evec);
#define CL_EM(NB_, V_) do { if (a.nt) hipLaunchKernelGGL((cl_step_envmajor_kernel<NB_, true, V_>), egrid, dim3(256 / V_), 0, s, a); \
                            else hipLaunchKernelGGL((cl_step_envmajor_kernel<NB_, false, V_>), egrid, dim3(256 / V_), 0, s, a); } while (0)
        if (enb == 17) { if (evec == 2) CL_EM(17, 2); else CL_EM(17, 1); }
        else { if (evec == 2) CL_EM(20, 2); else CL_EM(20, 1); }
#undef CL_EM
    } else if (lean_shape) {
        // one workgroup per CU at most: with more rounds the generic kernel's smaller register file (52 vs 88 VGPRs, two
        // workgroups per CU) wins again -- 17 x 262 144: 30.8 us vs 33.0 us
        switch (vec) {                                   // latency-ordered lean kernel (two buildings per wave at most)
#define CL_LEAN_CASE(V) case V: \
            name_add(tun, "%s<" #V ", %s%s>", kpi_lean ? "cl_step_lean_kpi_kernel" : (of && rkind_host != CLR_MARL) ? "cl_step_lean_obs_kernel" : "cl_step_lean_kernel", \
                     kpi_lean || (of && rkind_host != CLR_MARL) ? "" : "false, ", a.nt ? "true" : "false"); \
            if (kpi_lean) { \
                const size_t lds_k = lds + CL_OBS_FUSED_BLDG * sizeof(float);      /* + one baseline value per building */ \
                if (a.nt) hipLaunchKernelGGL((cl_step_lean_kpi_kernel<V, true>), grid, block, lds_k, s, a); \
                else hipLaunchKernelGGL((cl_step_lean_kpi_kernel<V, false>), grid, block, lds_k, s, a); \
            } else if (of && rkind_host != CLR_MARL) {       /* (MARL's reward plane is finished after the sweep the tile is filled in) */ \
                const size_t lds_o = lds + (size_t)tile * of->pitch * sizeof(float); \
                if (a.nt) hipLaunchKernelGGL((cl_step_lean_obs_kernel<V, true>), grid, block, lds_o, s, a, *of); \
                else hipLaunchKernelGGL((cl_step_lean_obs_kernel<V, false>), grid, block, lds_o, s, a, *of); \
                *fused = true; \
            } else if (const int rc = cl_tu_launch_lean(V, a.nt, grid.x, grid.y, block.x, lds, stream, &a)) \
                return hip_fail((hipError_t)rc, "cl_step_lean_kernel launch"); \
            break;
        CL_LEAN_CASE(1) CL_LEAN_CASE(2) CL_LEAN_CASE(4)
#undef CL_LEAN_CASE
        default: return fail(CL_EINVAL, "bad vec %d", vec);
        }
    } else if (a.n_chunks > 1 && tun.finish != 2 && (vec == 1 || vec == 2 || vec == 4) && !(tun.lean_variant & 16)) {
        // building-chunked battery + PV districts: the latency-ordered chunk kernel (lean_variant & 16 keeps cl_step_kernel: tests, A/B)
        a.fused_finish = can_defer ? 2 : 0;
        name_add(tun, "cl_step_lean_chunk_kernel<%d, %s, %s, 0>", vec, a.nt ? "true" : "false", can_defer ? "true" : "false");
        if (int rc = launch_lean_chunk<0>(vec, a.nt, can_defer, grid, block, lds, s, a)) return rc;
    } else if (a.n_chunks > 1 && (tun.finish == 2 || can_defer) && (vec == 1 || vec == 4)) {
        // building-chunked battery + PV districts (C4 with the 2022 device set): the instantiations that fold the chunk sums themselves
        // (finish = 2: their own, inside the launch; finish = 3: the previous step's, deferred)
        a.fused_finish = tun.finish == 2 ? 1 : 2;
        name_add(tun, "cl_step_kernel<%d, false, false, false, 0, true>", vec);
        // (four envs per lane x 16 waves: 64 KB of reduction rows + the 4 KB exchange tile of the deferred fold -- more dynamic LDS than a
        //  kernel gets without opting in where the runtime enforces the 64 KB default; gfx950's 160 KB hold it)
        if (lds > 64 * 1024) {
            const void* fn = vec == 1 ? reinterpret_cast<const void*>(cl_step_kernel<1, false, false, false, 0, true>)
                                      : reinterpret_cast<const void*>(cl_step_kernel<4, false, false, false, 0, true>);
            if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess)
                return hip_fail(e, "hipFuncSetAttribute(cl_step_kernel<.., FOLD>)");
        }
        const dim3 grid_sw = CL_SWAP_GRID ? dim3(grid.y, grid.x) : grid;      // (chunks along x: district_reduce's SWAP note)
        if (vec == 1) hipLaunchKernelGGL((cl_step_kernel<1, false, false, false, 0, true>), grid_sw, block, lds, s, a);
        else hipLaunchKernelGGL((cl_step_kernel<4, false, false, false, 0, true>), grid_sw, block, lds, s, a);
    } else {
        name_add(tun, "cl_step_kernel<%d, false, false, false, 0, false>", vec);
        switch (vec) {
        case 1: hipLaunchKernelGGL((cl_step_kernel<1, false, false>), grid, block, lds, s, a); break;
        case 2: hipLaunchKernelGGL((cl_step_kernel<2, false, false>), grid, block, lds, s, a); break;
        case 4: hipLaunchKernelGGL((cl_step_kernel<4, false, false>), grid, block, lds, s, a); break;
        default: return fail(CL_EINVAL, "bad vec %d", vec);
        }
    }
    if (a.n_chunks > 1) {
        if (!a.fused_finish) {
            name_add(tun, "cl_finish_kernel");
            hipLaunchKernelGGL(cl_finish_kernel, dim3((dims->n_env + 63) / 64, NQ), dim3(1024), 0, s, a, 0, (float*)nullptr);
        }
        if (((dims->flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT) == CLR_MARL) {
            const long long n = (long long)dims->n_env * dims->n_bldg;
            name_add(tun, "cl_marl_reward_kernel");
            hipLaunchKernelGGL(cl_marl_reward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
        }
    }
    if (dims->flags & CLD_KPI) {
        const long long n = (long long)dims->n_env * dims->n_bldg;
        // (without the detail planes -- lean districts -- the step launch above has updated every accumulator itself)
        if (kpi_full) {
            // (cl_step_full_kpi_kernel has updated every accumulator)
        } else if ((dims->flags & CLD_WRITE_DETAIL) && tun.kpi_passes == 2) {            // the round-1 / round-2 form: two passes (tests, A/B)
            name_add(tun, "cl_kpi_bldg_kernel+cl_kpi_env_kernel");
            hipLaunchKernelGGL(cl_kpi_bldg_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
            hipLaunchKernelGGL(cl_kpi_env_kernel, dim3((dims->n_env + 63) / 64), dim3(1024), 0, s, a);
        } else if (dims->flags & CLD_WRITE_DETAIL) {
            name_add(tun, "cl_kpi_kernel");
            hipLaunchKernelGGL(cl_kpi_kernel, dim3((dims->n_env + 63) / 64), dim3(1024), 0, s, a);
        }
    }
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_fail(e, "cl_step_kernel launch");
    return CL_OK;
}

int cl_finish_f32(const cl_dims* dims, float* out_bldg, float* out_env, int32_t t, void* stream) {
    if (int rc = check_dims(dims)) return rc;
    if (int rc = check_ptr(out_bldg, "out_bldg")) return rc;
    if (int rc = check_ptr(out_env, "out_env")) return rc;
    if (t < 0 || t >= dims->n_steps) return fail(CL_ERANGE, "t=%d outside [0, %d)", t, dims->n_steps);
    if (dims->n_bldg <= 32) return CL_OK;              // never building-chunked: every step launch finishes its own district sums
    if (int rc = no_pitch(dims, "cl_finish_f32")) return rc;
    StepArgs a = {};
    a.out_bldg = out_bldg; a.out_env = out_env; a.n_env = dims->n_env; a.n_bldg = dims->n_bldg; a.n_steps = dims->n_steps;
    a.flags = dims->flags; a.t = t; a.n_chunks = 0;    // (the kernel reads the chunk count next to the marker)
    hipLaunchKernelGGL(cl_finish_kernel, dim3((dims->n_env + 63) / 64, NQ), dim3(1024), 0, (hipStream_t)stream, a, 1, (float*)nullptr);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_fail(e, "cl_finish_kernel launch");
    return CL_OK;
}

int cl_step_flex_f32(const cl_dims* dims, const uint32_t* params, const float* ts, float* state, const float* actions,
                     int64_t act_stride_col, int64_t act_stride_env, float* out_bldg, float* out_env, float* kpi_bldg,
                     float* kpi_env, const cl_flex* flex, int32_t t, void* stream) {
    return step_impl(dims, params, ts, state, actions, act_stride_col, act_stride_env, out_bldg, out_env, kpi_bldg, kpi_env, flex, t, stream,
                     nullptr, nullptr);
}

int cl_step_observe_f32(const cl_dims* dims, const uint32_t* params, const float* ts, float* state, const float* actions,
                        int64_t act_stride_col, int64_t act_stride_env, float* out_bldg, float* out_env, float* kpi_bldg, float* kpi_env,
                        int32_t t, const float* obs_table, const int32_t* col_src, const float* col_scale, const cl_obs_dep* deps,
                        int32_t n_deps, float* obs, int32_t n_cols, int32_t obs_pitch, int32_t n_rows, int32_t obs_row, void* stream) {
    if (int rc = check_dims(dims)) return rc;
    if (int rc = check_ptr(obs_table, "obs_table")) return rc;
    if (int rc = check_ptr(obs, "obs")) return rc;
    if (!deps || n_deps != n_cols || n_cols <= 0 || n_cols > CLOB_MAX_DEPS)
        return fail(CL_EINVAL, "cl_step_observe_f32 takes the compact form: every one of the n_cols <= %d columns env-dependent and listed (n_cols=%d, n_deps=%d)",
                    CLOB_MAX_DEPS, n_cols, n_deps);
    if (obs_pitch < n_cols) return fail(CL_EINVAL, "obs_pitch=%d < n_cols=%d", obs_pitch, n_cols);
    if (obs_row < 0 || obs_row >= n_rows) return fail(CL_ERANGE, "obs_row=%d outside [0, %d)", obs_row, n_rows);
    // what the step launch itself can write: columns fed by the planes a lean wave holds in registers, rows padded to 16 bytes
    ObsFusedArgs of;
    bool can = obs_pitch == ((n_cols + 3) & ~3) && dims->n_bldg <= CL_OBS_FUSED_BLDG && tuning_of(dims).obs_variant == 0;
    bool lean_planes = true;                        // only what a battery + PV wave holds: battery state, net, reward
    int n = 0;
    for (int b = 0; b < dims->n_bldg && can; ++b) {
        of.start[b] = (short)n;
        for (int d = 0; d < n_deps; ++d) {
            const int src = deps[d].src, kind = src >> 28, pl = (src >> 20) & 0xFF;
            if (deps[d].col < 0 || deps[d].col >= n_cols || src < 0) return fail(CL_EINVAL, "deps[%d]: col=%d src=%d", d, deps[d].col, src);
            // (under CLD_F64_CHAIN the degraded-capacity plane carries the capacity LOSS: not an observation source -- round-5 advisor)
            if (kind == CLOB_KIND_STATE && pl == CLS_B_DEGCAP && (dims->flags & CLD_F64_CHAIN))
                return fail(CL_EINVAL, "deps[%d]: CLS_B_DEGCAP holds capacity - degraded_capacity under CLD_F64_CHAIN and cannot feed an observation column", d);
            if ((src & 0xFFFFF) != b) continue;
            const bool lean_pl = (kind == CLOB_KIND_STATE && (pl == CLS_B_SOC || pl == CLS_B_EFF || pl == CLS_B_DEGCAP)) ||
                                 (kind == CLOB_KIND_OUT && (pl == CLO_NET || pl == CLO_REWARD));
            lean_planes = lean_planes && lean_pl;
            can = can && (lean_pl || (kind == CLOB_KIND_STATE && (pl == CLS_CS_SOC || pl == CLS_HS_SOC || pl == CLS_DS_SOC)));      // + the tank planes of the thermal kernels
            of.deps[n++] = deps[d];
        }
    }
    can = can && n == n_deps;                       // (a building index beyond n_bldg would have been skipped)
    bool lean_ok = can && lean_planes;
    for (int b = 0; b < dims->n_bldg && lean_ok; ++b) lean_ok = (b + 1 < dims->n_bldg ? of.start[b + 1] : n) - of.start[b] <= CL_OBS_FUSED_PER_BLDG;
    of.lean_ok = lean_ok ? 1 : 0;
    for (int b = dims->n_bldg; b <= CL_OBS_FUSED_BLDG; ++b) of.start[b] = (short)n;
    of.obs = obs; of.table = obs_table; of.n_cols = n_cols; of.pitch = obs_pitch; of.row = obs_row;
    bool fused = false;
    if (int rc = step_impl(dims, params, ts, state, actions, act_stride_col, act_stride_env, out_bldg, out_env, kpi_bldg, kpi_env, nullptr, t,
                           stream, can ? &of : nullptr, &fused)) return rc;
    if (fused) return CL_OK;
    return cl_observe_f32(dims, obs_table, col_src, col_scale, deps, n_deps, state, out_bldg, nullptr, nullptr, 0, obs, n_cols, obs_pitch,
                          n_rows, obs_row, 0u, stream);
}

int cl_rollout_seq_f32(const cl_dims* dims, const uint32_t* params, const float* ts, float* state, const float* actions,
                       int64_t act_stride_step, int64_t act_stride_col, int64_t act_stride_env, const float* act_low,
                       const float* act_high, uint64_t seed, float* policy_actions, float* out_bldg, float* out_env,
                       float* ret_env, float* kpi_bldg, float* kpi_env, const cl_flex* flex, int32_t t0, int32_t k_steps, void* stream) {
    if (int rc = check_dims(dims)) return rc;
    if (int rc = check_ptr(actions, "actions", false)) return rc;
    if (int rc = check_ptr(ret_env, "ret_env", false)) return rc;
    if (!actions && dims->n_act_cols > 0) {
        if (!act_low || !act_high) return fail(CL_ENULL, "act_low / act_high are required for the on-device policy");
        if (int rc = check_ptr(policy_actions, "policy_actions")) return rc;
        if (dims->n_env % 4) return fail(CL_EALIGN, "the on-device policy needs n_env to be a multiple of 4 (got %d)", dims->n_env);
    }
    if (k_steps < 0 || t0 < 0 || t0 + k_steps > dims->n_steps)
        return fail(CL_ERANGE, "steps [%d, %d) outside [0, %d)", t0, t0 + k_steps, dims->n_steps);
    hipStream_t s = (hipStream_t)stream;
    const unsigned gx = (unsigned)((dims->n_env + 255) / 256);
    for (int k = 0; k < k_steps; ++k) {
        const int t = t0 + k;
        const float* a = actions ? actions + (long long)k * act_stride_step
                                 : policy_actions + (long long)(t & 3) * dims->n_act_cols * dims->n_env;
        if (!actions && dims->n_act_cols > 0 && (k == 0 || (t & 3) == 0))
            hipLaunchKernelGGL(cl_policy_kernel, dim3(gx, (unsigned)dims->n_act_cols), dim3(256), 0, s, policy_actions, act_low, act_high,
                               (unsigned long long)seed, dims->n_env, dims->n_act_cols, t >> 2, (unsigned)dims->env_offset);
        if (int rc = cl_step_flex_f32(dims, params, ts, state, a, actions ? act_stride_col : (int64_t)dims->n_env,
                                      actions ? act_stride_env : (int64_t)1, out_bldg, out_env, kpi_bldg, kpi_env, flex, t, stream))
            return rc;
        if (ret_env) {
            // (a deferred finish -- cl_tuning.finish = 3 -- would leave the previous step's reward here: fold now; a no-op where the step did not defer)
            if (tuning_of(dims).finish == 3)
                if (int rc = cl_finish_f32(dims, out_bldg, out_env, t, stream)) return rc;
            hipLaunchKernelGGL(cl_return_kernel, dim3(gx), dim3(256), 0, s, ret_env, out_env + (long long)CLQ_REWARD * dims->n_env, dims->n_env);
        }
    }
    // end of the sequence: out_env holds the last step's district sums whatever the finish mode
    if (!ret_env && k_steps > 0 && tuning_of(dims).finish == 3)
        if (int rc = cl_finish_f32(dims, out_bldg, out_env, t0 + k_steps - 1, stream)) return rc;
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_fail(e, "cl_rollout_seq_f32 launch");
    return CL_OK;
}

int cl_rollout_f32(const cl_dims* dims, const uint32_t* params, const float* ts, float* state, const float* actions,
                   int64_t act_stride_step, int64_t act_stride_col, int64_t act_stride_env, const float* act_low,
                   const float* act_high, uint64_t seed, float* out_bldg, float* out_env, float* ret_env,
                   int32_t t0, int32_t k_steps, void* stream) {
    if (int rc = check_dims(dims)) return rc;
    const cl_tuning& tun = tuning_of(dims);
    if (int rc = check_ptr(params, "params")) return rc;
    if (int rc = check_ptr(ts, "ts")) return rc;
    if (int rc = check_ptr(state, "state")) return rc;
    if (int rc = check_ptr(out_bldg, "out_bldg")) return rc;
    if (int rc = check_ptr(out_env, "out_env")) return rc;
    if (int rc = check_ptr(actions, "actions", false)) return rc;
    if (int rc = check_ptr(ret_env, "ret_env", false)) return rc;
    if (!actions && dims->n_act_cols > 0) {
        if (!act_low || !act_high) return fail(CL_ENULL, "act_low / act_high are required for the on-device policy");
    }
    if (((dims->flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT) == CLR_EV)
        return fail(CL_EINVAL, "reward kind CLR_EV needs the flexible-load tables (cl_rollout_seq_f32)");
    if (dims->flags & CLD_KPI) return fail(CL_EINVAL, "the fused rollout keeps no streaming KPIs: use cl_rollout_seq_f32 with CLD_KPI");
    if (dims->flags & CLD_F64_MAPS) return fail(CL_EINVAL, "the fused rollout evaluates the battery map in fp32 or as the float64 chain (CLD_F64_CHAIN): use cl_rollout_seq_f32 with CLD_F64_MAPS");
    if (k_steps < 0 || t0 < 0 || t0 + k_steps > dims->n_steps)
        return fail(CL_ERANGE, "steps [%d, %d) outside [0, %d)", t0, t0 + k_steps, dims->n_steps);
    if (actions && act_stride_env == 1 && ((act_stride_col % 4) != 0 || (act_stride_step % 4) != 0))
        return fail(CL_EALIGN, "action strides must be multiples of 4 floats for the coalesced layout");

    RolloutArgs r;
    StepArgs& a = r.s;
    a.params = params; a.ts = ts; a.state = state; a.actions = actions; a.out_bldg = out_bldg; a.out_env = out_env;
    a.kpi_bldg = nullptr; a.kpi_env = nullptr;
    a.act_stride_col = act_stride_col; a.act_stride_env = act_stride_env;
    a.n_env = dims->n_env; a.n_bldg = dims->n_bldg; a.n_steps = dims->n_steps;
    a.flags = dims->flags; a.t = t0; a.b_chunk = dims->n_bldg; a.n_chunks = 1; a.env_row0 = dims->env_row0; a.env_offset = (unsigned)dims->env_offset;
    a.nt = 0; a.fused_finish = 0;
    a.ld = pitch_of(dims);
    if (a.ld != a.n_env && (!(dims->flags & CLD_LEAN) || (dims->flags & CLD_WRITE_DETAIL) || dims->n_bldg > 32))
        return fail(CL_EINVAL, "env_pitch=%d != n_env=%d: the fused rollout implements a pitch for CLD_LEAN districts of up to 32 buildings", a.ld, a.n_env);
    r.act_stride_step = act_stride_step; r.act_low = act_low; r.act_high = act_high; r.ret_env = ret_env; r.seed = seed;
    r.t0 = t0; r.k_steps = k_steps;
    const bool full = !(dims->flags & CLD_LEAN) || (dims->flags & CLD_WRITE_DETAIL);
    // a wave keeps the state of MB buildings in registers: two battery + PV buildings, one thermal one.  Districts beyond 16 MB buildings are
    // cut into chunks of `b_chunk` <= 16 MB along gridDim.y (cl_rollout_kernel's note); cl_finish_kernel folds the chunk sums once per launch
    const int mb_max = full ? 1 : 2;
    const bool chunked = dims->n_bldg > 16 * mb_max;
    int mb = full ? 1 : (dims->n_bldg > 16 || (dims->flags & CLD_F64_CHAIN) ? 2 : 1);      // (the one-building chain instantiation reserves scratch memory)
    if (chunked) {
        if (((dims->flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT) == CLR_MARL)
            return fail(CL_EINVAL, "the fused rollout of a building-chunked district (n_bldg=%d) cannot couple the buildings inside a step (MARL): use cl_rollout_seq_f32", dims->n_bldg);
        // (the packed thermal kernel: EIGHT buildings = waves per workgroup -- three / two workgroups per CU instead of one sixteen-wave one at its
        //  84 / 97 registers and 50 / 76 KB of LDS: 1024 buildings x 1024 / 2048 / 8192 envs 289 -> 263 / 557 -> 498 / 2173 -> 1873 us per 24 steps
        //  under the float64 chain, 195 -> 183 / 378 -> 336 / 1458 -> 1258 us on the fp32 map; profiles/r06m_*)
        const bool packed_shape = full && !(dims->flags & CLD_WRITE_DETAIL) && tun.full_variant != 1 && dims->n_act_cols <= 65536;
        a.b_chunk = tun.b_chunk > 0 ? tun.b_chunk : packed_shape ? 8 : 16 * mb_max;
        if (a.b_chunk > 16 * mb_max) return fail(CL_EINVAL, "b_chunk=%d: a fused-rollout workgroup holds at most %d buildings of this district", a.b_chunk, 16 * mb_max);
        mb = mb_max;                      // (battery + PV: always two buildings per wave -- the one-building chunked instantiations spill)
        a.n_chunks = (dims->n_bldg + a.b_chunk - 1) / a.b_chunk;
        // the reserved plane holds n_chunks x NQ partial sums, one return row per chunk and the marker words
        // (and, in front of those, the ticket words of the one-step launches' in-launch fold: they stay zero between launches)
        if (((long long)a.n_chunks * (NQ + 1)) * dims->n_env + (dims->n_env + 63) / 64 + 4 > (long long)dims->n_bldg * dims->n_env)
            return fail(CL_EINVAL, "b_chunk=%d leaves no room for the %d chunk partial sums of the fused rollout", a.b_chunk, a.n_chunks);
    }
    a.nw = tun.nw ? tun.nw : ((chunked ? a.b_chunk : dims->n_bldg) + mb - 1) / mb;
    if (a.nw * mb < (chunked ? a.b_chunk : dims->n_bldg) || a.nw > 16) return fail(CL_EINVAL, "bad nw %d", a.nw);
    // two envs per lane where its 128-env workgroups come in (nearly) full rounds of one per CU: 17 x 32 768 (the C5 per-GPU shard) 2.60 ->
    // 2.31 us per step, 65 536 4.99 -> 4.19, 98 304 7.24 -> 6.19, 131 072 9.54 -> 8.21; 49 152 = 1.5 rounds: 4.20 vs 3.89 at one env per lane
    // (scripts/rollout_vec_probe.py); chunked districts: the workgroup count is env tiles x chunks
    const long long wg2 = (long long)((dims->n_env + 127) / 128) * a.n_chunks, rounds2 = (wg2 + 255) / 256;
    const bool full_rounds = (long long)dims->n_env * a.n_chunks >= 32768 && wg2 * 100 >= rounds2 * 256 * 85 && dims->n_env >= 128;
    const bool chain = dims->flags & CLD_F64_CHAIN;          // (the float64 soc chain: two envs per lane only where one workgroup row holds the district)
    int vec = tun.vec ? tun.vec : ((!full && (actions == nullptr || act_stride_env == 1) && full_rounds) ? 2 : 1);
    if (chain && (full || vec > 2)) vec = 1;                 // (chunked battery + PV districts keep the two-env pack under the chain too: 1024 x 1024 / x 8192 127.6 -> 107.9 / 932.9 -> 798.9 us per 24 steps, profiles/r06z_*)
    const int tile = 64 * vec;
    const unsigned grid = (unsigned)((dims->n_env + tile - 1) / tile);
    const size_t lds = (size_t)a.nw * NQ * tile * sizeof(float);
    const dim3 block(64 * a.nw);
    name_reset(tun);
    // Thermal / outage districts without detail planes (round 6): the pack-generic unit of cl_full.h inside the K-step loop (cl_rollout_full_kernel),
    // two envs per lane where the battery map is fp32 and the actions are contiguous along the envs; cl_tuning.full_variant = 1 keeps the scalar unit
    const bool packed = full && mb == 1 && !(dims->flags & CLD_WRITE_DETAIL) && tun.full_variant != 1 && dims->n_act_cols <= 65536;      // (column ids travel 16 bits each)
    if (packed) {
        const bool marl = ((dims->flags & CLD_REWARD_MASK) >> CLD_REWARD_SHIFT) == CLR_MARL;      // (never chunked: refused above; one env per lane -- the two-env MARL instantiation parks 36 bytes in scratch)
        const int fvec = (chain || marl || (actions && act_stride_env != 1) || tun.vec == 1) ? 1 : 2;
        const dim3 pgrid((unsigned)((dims->n_env + 64 * fvec - 1) / (64 * fvec)), (unsigned)a.n_chunks);
        // LDS: the reduction rows, or MARL's exchange row + every wave's Philox blocks (cl_rollout_full_kernel's note) -- 100 / 152 KB at 16 waves
        const size_t tile_b = (size_t)64 * fvec * sizeof(float);
        const size_t rows = (size_t)a.nw * NQ * tile_b, rnd = (size_t)a.nw * tile_b + (size_t)a.nw * (fvec == 1 ? CL_ROLLOUT_RND_ROWS<1> : CL_ROLLOUT_RND_ROWS<2>) * tile_b;
        const size_t plds = rows > rnd ? rows : rnd;
        name_add(tun, "cl_rollout_full_kernel<%d, %s, %d, %s>", fvec, chunked ? "true" : "false", chain ? 2 : 0, marl ? "true" : "false");
#define CL_RFM(V, C, P, M) do { \
            if (plds > 64 * 1024) if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(cl_rollout_full_kernel<V, C, P, M>), plds); e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(cl_rollout_full_kernel)"); \
            hipLaunchKernelGGL((cl_rollout_full_kernel<V, C, P, M>), pgrid, block, plds, (hipStream_t)stream, r); } while (0)
#define CL_RF(V, C, P) do { if (marl && !C) CL_RFM(V, false, P, true); else CL_RFM(V, C, P, false); } while (0)
        if (chain) { if (chunked) CL_RF(1, true, 2); else CL_RF(1, false, 2); }
        else if (fvec == 2) { if (chunked) CL_RFM(2, true, 0, false); else CL_RFM(2, false, 0, false); }
        else { if (chunked) CL_RF(1, true, 0); else CL_RF(1, false, 0); }
#undef CL_RFM
#undef CL_RF
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_fail(e, "cl_rollout_full_kernel launch");
    } else {
    const int key = (full ? 100 : 0) + vec * 10 + mb;
    if ((key != 11 && key != 12 && key != 21 && key != 22 && key != 111) || ((chunked || chain) && key != 12 && key != 22 && key != 111) )
        return fail(CL_EINVAL, "no rollout kernel for vec %d / buildings-per-wave %d / %s", vec, mb, full ? "full" : "lean");
    const bool pin = chunked || chain || (long long)grid * a.nw > 5 * 1024;
    // (spelled as rocprofv3 prints them: every template argument, defaults included)
    name_add(tun, "cl_rollout_kernel<%d, %s, %d, %s, %s, %d>", vec, full ? "true" : "false", mb, (key == 12 && !pin) ? "false" : "true", chunked ? "true" : "false", chain ? 2 : 0);
    const int rc = cl_tu_launch_rollout(key + (chunked ? 1000 : 0) + (chain ? 2000 : 0), pin, grid, (unsigned)a.n_chunks, block.x, lds, stream, &r);
    if (rc) return hip_fail((hipError_t)rc, "cl_rollout_kernel launch");
    }
    if (chunked && k_steps > 0) {
        // the last step's district sums (and the K-step return): one fold per launch
        StepArgs f = a;
        f.t = t0 + k_steps - 1;
        name_add(tun, "cl_finish_kernel");
        hipLaunchKernelGGL(cl_finish_kernel, dim3((dims->n_env + 63) / 64, NQ + (ret_env ? 1 : 0)), dim3(1024), 0, (hipStream_t)stream, f, 0, ret_env);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_fail(e, "cl_finish_kernel launch");
    }
    return CL_OK;
}

int cl_lstm_reset_f32(const cl_dims* dims, float* hist, float* hidden, float* kpi_comfort, void* stream) {
    if (int rc = check_dims(dims)) return rc;
    if (int rc = no_pitch(dims, "cl_lstm_reset_f32")) return rc;
    if (int rc = check_ptr(hist, "hist")) return rc;
    if (int rc = check_ptr(hidden, "hidden")) return rc;
    const size_t plane = (size_t)dims->n_env * dims->n_bldg * sizeof(float);
    if (hipError_t e = hipMemsetAsync(hist, 0, CL_LSTM_NHIST * plane, (hipStream_t)stream); e != hipSuccess) return hip_fail(e, "memset hist");
    if (hipError_t e = hipMemsetAsync(hidden, 0, CL_LSTM_NHIDDEN * plane, (hipStream_t)stream); e != hipSuccess) return hip_fail(e, "memset hidden");
    if (kpi_comfort) {
        if (int rc = check_ptr(kpi_comfort, "kpi_comfort")) return rc;
        const long long units = (long long)dims->n_env * dims->n_bldg;
        hipLaunchKernelGGL(cl_kpi_comfort_reset_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kpi_comfort, units);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_fail(e, "cl_kpi_comfort_reset_kernel launch");
    }
    return CL_OK;
}

int cl_lstm_step_f32(const cl_dims* dims, const float* lstm_w, const uint16_t* lstm_wb, const float* dyn_pre, const float* cool_dem,
                     const float* heat_dem, float* hist, float* hidden, float* indoor_temp, float* comfort,
                     float* kpi_comfort, int32_t t, void* stream) {
    if (int rc = check_dims(dims)) return rc;
    if (int rc = no_pitch(dims, "cl_lstm_step_f32")) return rc;
    const cl_tuning& tun = tuning_of(dims);
    if (int rc = check_ptr(lstm_w, "lstm_w")) return rc;
    if (int rc = check_ptr(dyn_pre, "dyn_pre")) return rc;
    if (int rc = check_ptr(cool_dem, "cool_dem")) return rc;
    if (int rc = check_ptr(hist, "hist")) return rc;
    if (int rc = check_ptr(hidden, "hidden")) return rc;
    if (int rc = check_ptr(indoor_temp, "indoor_temp")) return rc;
    if (t < 0 || t >= dims->n_steps) return fail(CL_ERANGE, "t=%d outside [0, %d)", t, dims->n_steps);
    LstmArgs a;
    a.lstm_w = lstm_w; a.dyn_pre = dyn_pre; a.cool_dem = cool_dem; a.hist = hist; a.hidden = hidden; a.indoor_temp = indoor_temp;
    a.heat_dem = heat_dem; a.comfort = comfort; a.kpi_comfort = kpi_comfort;
    if (int rc = check_ptr(kpi_comfort, "kpi_comfort", false)) return rc;
    a.n_env = dims->n_env; a.n_bldg = dims->n_bldg; a.t = t; a.env_row0 = dims->env_row0;
    const dim3 grid((dims->n_env + 127) / 128, dims->n_bldg);          // 4 waves x 32 envs per workgroup
    a.lstm_wb = lstm_wb;
    if (int rc = check_ptr(lstm_wb, "lstm_wb", false)) return rc;
    const bool f16 = dims->flags & CLD_LSTM_F16;
    const bool two = dims->flags & CLD_LSTM_TWO_DEMANDS;
    if (two && !heat_dem) return fail(CL_EINVAL, "CLD_LSTM_TWO_DEMANDS needs heat_dem");
    name_reset(tun);
#define CL_LSTM_LAUNCH(DBG, SPLIT) do { if (two) { name_add(tun, "cl_lstm_kernel<" #DBG ", " #SPLIT ", true>"); \
    hipLaunchKernelGGL((cl_lstm_kernel<DBG, SPLIT, true>), grid, dim3(256), 0, (hipStream_t)stream, a); } else { name_add(tun, "cl_lstm_kernel<" #DBG ", " #SPLIT ", false>"); \
    hipLaunchKernelGGL((cl_lstm_kernel<DBG, SPLIT, false>), grid, dim3(256), 0, (hipStream_t)stream, a); } } while (0)
#define CL_LSTM_LAUNCH_WB(DBG) do { if (f16) CL_LSTM_LAUNCH(DBG, 2); else CL_LSTM_LAUNCH(DBG, 1); } while (0)
    switch (tun.lstm_variant) {            // timing experiments (scripts/lstm_check.py); 0 in production
    case 1: CL_LSTM_LAUNCH(1, 0); break;
    case 2: CL_LSTM_LAUNCH(2, 0); break;
    case 3: CL_LSTM_LAUNCH(0, 0); break;                                   // f32 MFMA path
    case 5: if (!lstm_wb) return fail(CL_EINVAL, "lstm_variant 5 needs lstm_wb"); CL_LSTM_LAUNCH_WB(1); break;
    case 6: if (!lstm_wb) return fail(CL_EINVAL, "lstm_variant 6 needs lstm_wb"); CL_LSTM_LAUNCH_WB(2); break;
    case 32: if (!lstm_wb) return fail(CL_EINVAL, "lstm_variant 32 needs lstm_wb"); CL_LSTM_LAUNCH_WB(32); break;   // common-denominator cell update (experiment)
    case 8: if (!lstm_wb || f16) return fail(CL_EINVAL, "lstm_variant 8 needs bf16 lstm_wb"); CL_LSTM_LAUNCH(8, 1); break;   // two-term bf16 split
    default:
        if (lstm_wb) CL_LSTM_LAUNCH_WB(0);
        else CL_LSTM_LAUNCH(0, 0);
    }
#undef CL_LSTM_LAUNCH_WB
#undef CL_LSTM_LAUNCH
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_fail(e, "cl_lstm_kernel launch");
    return CL_OK;
}

int cl_lstm_generic_step_f32(const cl_dims* dims, const float* lstm_w, const float* dyn_pre, const float* gen_w, int64_t gen_w_stride,
                             const float* gen_pre, float* gen_hidden, int32_t gen_h, int32_t gen_layers, const float* cool_dem, const float* heat_dem,
                             float* hist, float* indoor_temp, float* comfort, float* kpi_comfort, int32_t t, void* stream) {
    if (int rc = check_dims(dims)) return rc;
    if (int rc = no_pitch(dims, "cl_lstm_generic_step_f32")) return rc;
    if (int rc = check_ptr(lstm_w, "lstm_w")) return rc;
    if (int rc = check_ptr(dyn_pre, "dyn_pre")) return rc;
    if (int rc = check_ptr(gen_w, "gen_w")) return rc;
    if (int rc = check_ptr(gen_pre, "gen_pre")) return rc;
    if (int rc = check_ptr(gen_hidden, "gen_hidden")) return rc;
    if (int rc = check_ptr(cool_dem, "cool_dem")) return rc;
    if (int rc = check_ptr(hist, "hist")) return rc;
    if (int rc = check_ptr(indoor_temp, "indoor_temp")) return rc;
    if (int rc = check_ptr(kpi_comfort, "kpi_comfort", false)) return rc;
    if (t < 0 || t >= dims->n_steps) return fail(CL_ERANGE, "t=%d outside [0, %d)", t, dims->n_steps);
    if (gen_h < 1 || gen_h > CL_LSTM_GEN_HMAX) return fail(CL_EINVAL, "gen_h=%d outside [1, %d]", gen_h, CL_LSTM_GEN_HMAX);
    if (gen_layers != 1 && gen_layers != 2) return fail(CL_EINVAL, "gen_layers=%d (1 or 2: the deepest model among the generic-kernel buildings)", gen_layers);
    const long long need = (long long)gen_h * 12 + 3ll * gen_h * gen_h * 4 + gen_h * 4 + gen_h;
    if (gen_w_stride < need) return fail(CL_EINVAL, "gen_w_stride=%lld < %lld floats for hidden size %d", (long long)gen_w_stride, need, gen_h);
    LstmGenArgs g;
    LstmArgs& a = g.s;
    a.lstm_w = lstm_w; a.lstm_wb = nullptr; a.dyn_pre = dyn_pre; a.cool_dem = cool_dem; a.hist = hist; a.hidden = nullptr;
    a.indoor_temp = indoor_temp; a.heat_dem = heat_dem; a.comfort = comfort; a.kpi_comfort = kpi_comfort;
    a.n_env = dims->n_env; a.n_bldg = dims->n_bldg; a.t = t; a.env_row0 = dims->env_row0;
    g.gen_w = gen_w; g.gen_pre = gen_pre; g.gen_hidden = gen_hidden; g.H = gen_h; g.gw = gen_w_stride;
    // LDS: hidden states [layers][2][H][64] and, when they fit beside them, the recurrent matrices (WHH0 (, WIH1, WHH1): [H][H][4] each),
    // sized for the deepest model of the district (`gen_layers`)
    const size_t lds_h = (size_t)gen_layers * 2 * gen_h * 64 * sizeof(float), lds_w = (size_t)(gen_layers == 2 ? 3 : 1) * gen_h * gen_h * 4 * sizeof(float);
    const bool staged = lds_h + lds_w <= 150 * 1024;
    const size_t lds = lds_h + (staged ? lds_w : 0);
    const void* fn = staged ? reinterpret_cast<const void*>(cl_lstm_generic_kernel<true>) : reinterpret_cast<const void*>(cl_lstm_generic_kernel<false>);
    if (lds > 64 * 1024) {
        if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess)
            return hip_fail(e, "hipFuncSetAttribute(cl_lstm_generic_kernel)");
    }
    const dim3 ggrid((dims->n_env + 63) / 64, dims->n_bldg), gblock(64 * CL_GEN_NWV);
    if (staged) hipLaunchKernelGGL(cl_lstm_generic_kernel<true>, ggrid, gblock, lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(cl_lstm_generic_kernel<false>, ggrid, gblock, lds, (hipStream_t)stream, g);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_fail(e, "cl_lstm_generic_kernel launch");
    return CL_OK;
}

int cl_observe_f32(const cl_dims* dims, const float* obs_table, const int32_t* col_src, const float* col_scale,
                   const cl_obs_dep* deps, int32_t n_deps, const float* state, const float* out_bldg, const float* indoor_temp,
                   const float* extra, int32_t n_extra_rows, float* obs, int32_t n_cols,
                   int32_t obs_pitch, int32_t n_rows, int32_t row, uint32_t flags, void* stream) {
    if (int rc = check_dims(dims)) return rc;
    const cl_tuning& tun = tuning_of(dims);
    if (int rc = check_ptr(obs_table, "obs_table")) return rc;
    if (int rc = check_ptr(obs, "obs")) return rc;
    if (n_cols <= 0 || n_rows <= 0) return fail(CL_EINVAL, "bad observation table shape [%d][%d]", n_rows, n_cols);
    if (obs_pitch < n_cols) return fail(CL_EINVAL, "obs_pitch=%d < n_cols=%d", obs_pitch, n_cols);
    if (row < 0 || row >= n_rows) return fail(CL_ERANGE, "row=%d outside [0, %d)", row, n_rows);
    const bool all_exo = (flags & CLOB_ALL_EXOGENOUS) != 0;
    if (!all_exo) {
        if (int rc = check_ptr(col_src, "col_src")) return rc;
        if (int rc = check_ptr(col_scale, "col_scale")) return rc;
        if (int rc = check_ptr(state, "state")) return rc;
        if (int rc = check_ptr(out_bldg, "out_bldg")) return rc;
        if (int rc = check_ptr(indoor_temp, "indoor_temp", false)) return rc;
    }
    ObsArgs a;
    a.row = obs_table + (size_t)row * n_cols; a.col_src = col_src; a.col_scale = col_scale; a.state = state;
    a.out_bldg = out_bldg; a.indoor_temp = indoor_temp; a.obs = obs; a.extra = extra; a.n_extra_rows = n_extra_rows;
    a.n_env = dims->n_env; a.n_bldg = dims->n_bldg; a.n_cols = n_cols; a.all_exo = all_exo ? 1 : 0;
    a.ld = pitch_of(dims);
    if (a.ld != a.n_env && (indoor_temp || extra)) return fail(CL_EINVAL, "cl_observe_f32: env_pitch=%d != n_env=%d is not implemented with LSTM / flexible-load planes", a.ld, a.n_env);
    a.env_row0 = dims->env_row0;
    const bool vec4 = obs_pitch % 4 == 0;            // 16-byte stores need 16-byte aligned rows
    a.pitch = obs_pitch;
    const int n_seg = (n_cols + OBS_SEG - 1) / OBS_SEG;
    const dim3 grid((dims->n_env + OBS_TILE - 1) / OBS_TILE, n_seg);
    const int padded = obs_pitch < ((n_cols + 3) & ~3) ? obs_pitch : ((n_cols + 3) & ~3);   // columns written per row
    a.padded = padded;
    const bool listed = all_exo || (deps != nullptr && n_deps >= 0 && n_deps <= OBS_DEP_MAX);
    if (listed && !all_exo)
        for (int d = 0; d < n_deps; ++d)
            if (deps[d].col < 0 || deps[d].col >= n_cols || deps[d].src < 0)
                return fail(CL_EINVAL, "deps[%d]: col=%d src=%d", d, deps[d].col, deps[d].src);
    // Kernel choice (measured on MI355X, scripts/observe_bench.py / profiles/):
    //  * narrow vectors (half a wave of 16-byte column groups per row or less): LDS-tile kernel (52 columns: 11 vs 15 us);
    //  * wide vectors when the launch runs in more than ~one round of resident waves (> 96k envs) and the host-side list
    //    leaves no lane with more than OBS_WSLOTS dependent columns: wave-independent kernel -- its waves never
    //    synchronise, so the read latency of starting waves hides behind the stores of running ones
    //    (262 144 x 476: 112 vs 120 us row-wise; 65 536 x 476, a single round: 30 vs 27 us);
    //  * everything else (no list, > 1024 columns, odd pitch, crowded lanes): row-wise kernel.
    ObsTileArgs t;
    if (n_seg == 1 && listed) {
        static_assert(OBS_DEP_MAX == CLOB_MAX_DEPS, "header and kernel disagree");
        t.n_deps = all_exo ? 0 : n_deps;
        for (int d = 0; d < t.n_deps; ++d) t.deps[d] = deps[d];
    }
    const bool narrow = tun.obs_variant == 2 || (tun.obs_variant == 0 && padded <= 128);
    bool wide = n_seg == 1 && listed && vec4 && (tun.obs_variant == 3 || (tun.obs_variant == 0 && !narrow && dims->n_env > 98304));
    if (wide) {
        int owned[256] = {0};
        for (int d = 0; d < t.n_deps && wide; ++d) wide = ++owned[(deps[d].col >> 2) & 63] <= OBS_WSLOTS;
    }
    // every column env-dependent (the compact observation form) and few of them: a plain transpose of the planes
    const bool transpose = !all_exo && n_seg == 1 && listed && vec4 && t.n_deps == n_cols && n_cols <= OBS_TCOLS && obs_pitch <= OBS_TCOLS + 4 &&
                           (tun.obs_variant == 0 || tun.obs_variant == 4);
    // wide vectors in one segment with a host-side list: the one-round-trip row-wise kernel (obs_variant 5 forces it, 1 keeps the round-1 one)
    int row1_slots = 0;                               // most dependent columns any lane owns (lane = 16-byte column group mod 64)
    if (n_seg == 1 && listed) {
        int owned[64] = {0};
        for (int d = 0; d < t.n_deps; ++d) { const int n = ++owned[(deps[d].col >> 2) & 63]; row1_slots = n > row1_slots ? n : row1_slots; }
    }
    // (profiles/r04_observe_bench.log, 65 536 envs, round-1 kernel -> this one: 476 columns 27.6 -> 26.6 us, 527: 32.7 -> 31.5, 272: 26.9 -> 23.4,
    //  245: 19.6 -> 20.6 -- kept on the round-1 kernel --, 262 144 x 476: 122 -> 110 us, where it also beats the wave-independent kernel's 113.5)
    const bool row1 = n_seg == 1 && listed && vec4 && !transpose && row1_slots <= 4 && (tun.obs_variant == 5 || (tun.obs_variant == 0 && !narrow && padded >= 256));
    if (row1) wide = false;
    if (transpose) {
        t.o = a;
        hipLaunchKernelGGL(cl_observe_transpose_kernel, dim3((dims->n_env + OBS_TILE - 1) / OBS_TILE), dim3(OBS_THREADS), 0, (hipStream_t)stream, t);
    } else if (wide) {
        t.o = a;
        const int n_waves = (dims->n_env + OBS_WROWS - 1) / OBS_WROWS, per_wg = OBS_THREADS / 64;
        const dim3 wgrid((n_waves + per_wg - 1) / per_wg);
        if (padded <= 512) hipLaunchKernelGGL(cl_observe_wave_kernel<2>, wgrid, dim3(OBS_THREADS), 0, (hipStream_t)stream, t);
        else hipLaunchKernelGGL(cl_observe_wave_kernel<4>, wgrid, dim3(OBS_THREADS), 0, (hipStream_t)stream, t);
    } else if (row1) {
        t.o = a;
        const dim3 rgrid((dims->n_env + OBS_TILE - 1) / OBS_TILE);
        if (padded <= 512) {
            if (row1_slots <= 2) hipLaunchKernelGGL((cl_observe_row1_kernel<2, 2>), rgrid, dim3(OBS_THREADS), 0, (hipStream_t)stream, t);
            else hipLaunchKernelGGL((cl_observe_row1_kernel<2, 4>), rgrid, dim3(OBS_THREADS), 0, (hipStream_t)stream, t);
        } else hipLaunchKernelGGL((cl_observe_row1_kernel<4, 4>), rgrid, dim3(OBS_THREADS), 0, (hipStream_t)stream, t);
    } else if (n_seg == 1 && padded == obs_pitch && listed && narrow && dims->env_row0 == nullptr) {   // one template row per launch
        int r = tun.obs_rows ? tun.obs_rows : 16;          // 16 envs = one 64-byte line of every dependent plane
        while (r * obs_pitch > OBS_BUF) r >>= 1;       // pitch <= OBS_SEG + 3 -> r >= 4
        a.sub_rows = r;
        t.o = a;
        const int n_blocks = (dims->n_env + r - 1) / r, per_wg = OBS_TILE / r;
        hipLaunchKernelGGL(cl_observe_tile_kernel, dim3((n_blocks + per_wg - 1) / per_wg), dim3(OBS_THREADS), 0, (hipStream_t)stream, t);
    } else if (vec4) hipLaunchKernelGGL(cl_observe_kernel<4>, grid, dim3(OBS_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(cl_observe_kernel<1>, grid, dim3(OBS_THREADS), 0, (hipStream_t)stream, a);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return hip_fail(e, "cl_observe_kernel launch");
    return CL_OK;
}

float cl_philox_uniform(uint64_t seed, uint32_t env, uint32_t col, uint32_t t) {
    return cl::philox_u01(seed, env, col, t);
}

}  // extern "C"
#endif  // CL_TU_NOSLP
