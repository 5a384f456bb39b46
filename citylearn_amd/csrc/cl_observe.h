// Observation epilogue: writes the Gym observation tensor obs[n_env][n_cols] (policy layout:
// one contiguous vector per environment) after a step.
//
// Reference: Building.observations / _get_observations_data (building.py:1115-1219, 1336-1481) called per building per
// step, CityLearnEnv.observations (citylearn.py:451-485), NormalizedObservationWrapper (wrappers.py:39-167).  ~93 % of
// the columns do not depend on the environment (calendar, weather, prices, loads ...): the host packs them -- already
// min-max / sin-cos normalised if wanted -- into `obs_table[row][col]`; env-dependent columns are an affine map of a
// device plane, obs = plane[b][env] * col_scale[col] + obs_table[row][col].
//
// The kernel is a pure HBM *write* stream (n_env * n_cols * 4 B; reads are the few dependent planes): a workgroup owns
// 64 consecutive envs x one segment of <= OBS_SEG columns; wave w writes rows w, w+4, ...  Dependent planes are staged
// through LDS (coalesced 256 B reads along the env axis, transposed on the way out).
#pragma once

namespace {

constexpr int OBS_TILE = 64;        // envs per workgroup (one lane each while staging planes)
constexpr int OBS_SEG = 1024;       // columns per segment (grid.y)
constexpr int OBS_DEP_MAX = 64;     // dependent columns staged through LDS per segment; the rest read HBM directly
constexpr int OBS_THREADS = 256;

struct ObsArgs {
    const float* __restrict__ row;          // obs_table + row * n_cols
    const int32_t* __restrict__ env_row0;   // per-env-block episode offsets (cl_dims.env_row0) or null: row += env_row0[block]
    const int32_t* __restrict__ col_src;    // [n_cols] -1: env-independent; else kind << 28 | plane << 20 | building
    const float* __restrict__ col_scale;    // [n_cols]
    const float* __restrict__ state;        // [CL_NS][B][E]
    const float* __restrict__ out_bldg;     // [CL_NO][B][E]
    const float* __restrict__ indoor_temp;  // [B][E] or null
    const float* __restrict__ extra;        // [planes][n_extra_rows][E] or null (CLOB_KIND_EXTRA)
    int n_extra_rows;
    float* __restrict__ obs;                // [E][pitch]
    int n_env, n_bldg, n_cols, pitch;
    int ld;                                 // row stride of the state / out_bldg planes (cl_dims.env_pitch; = n_env unless padded)
    int padded;                             // columns written per row: n_cols rounded up to 4, at most pitch
    int sub_rows;                           // tile kernel: rows per LDS sub-tile (power of two, 4..64)
    int all_exo;                            // reset observation: every column comes from the table
};

CL_DEV const float* obs_plane(const ObsArgs& a, int s) {
    const int kind = s >> 28, plane = (s >> 20) & 0xFF, b = s & 0xFFFFF;
    // (the LSTM stage's and the flexible loads' planes are never pitched: the host refuses a pitch for districts that have them)
    const long long pl = (long long)a.ld * a.n_bldg;
    const float* base = kind == 0 ? a.state + plane * pl : kind == 1 ? a.out_bldg + plane * pl
                      : kind == 2 ? a.indoor_temp : a.extra + (long long)plane * a.n_extra_rows * a.n_env;
    return base + (long long)b * (kind <= 1 ? a.ld : a.n_env);
}

// VEC = 4: row pitch is a multiple of 4 floats -> every lane owns fixed 16-byte column groups (its env-independent
// values live in registers for the whole tile) and a wave writes 1 KB contiguous per store instruction.
// VEC = 1: arbitrary pitch, one column per lane-slot.
template <int VEC>
__global__ __launch_bounds__(OBS_THREADS) void cl_observe_kernel(ObsArgs a) {
    constexpr int GPL = OBS_SEG / VEC / 64;         // column groups per lane
    __shared__ int slot_s[OBS_SEG];                 // -1 exogenous | LDS slot | >= OBS_DEP_MAX: direct global read
    __shared__ float dep_s[OBS_DEP_MAX][OBS_TILE + 1];
    __shared__ int dep_src_s[OBS_DEP_MAX];
    __shared__ float dep_scale_s[OBS_DEP_MAX], dep_base_s[OBS_DEP_MAX];
    __shared__ int n_dep_s;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int env0 = blockIdx.x * OBS_TILE;
    const int c0 = blockIdx.y * OBS_SEG;
    const int seg_n = min(OBS_SEG, a.n_cols - c0);                 // logical columns of this segment
    const int seg_w = min(OBS_SEG, a.padded - c0);                 // incl. the pad columns of the last segment
    const int n_rows = min(OBS_TILE, a.n_env - env0);
    // OBS_TILE divides CL_ROW0_BLOCK: the table row of this tile is workgroup-uniform
    const float* __restrict__ trow = a.row + (a.env_row0 ? (long long)a.env_row0[env0 / CL_ROW0_BLOCK] * a.n_cols : 0);
    if (tid == 0) n_dep_s = 0;
    __syncthreads();
    for (int c = tid; c < seg_n; c += OBS_THREADS) {
        const int s = a.all_exo ? -1 : a.col_src[c0 + c];
        int slot = -1;
        if (s >= 0) {
            slot = atomicAdd(&n_dep_s, 1);
            if (slot < OBS_DEP_MAX) { dep_src_s[slot] = s; dep_scale_s[slot] = a.col_scale[c0 + c]; dep_base_s[slot] = trow[c0 + c]; }
        }
        slot_s[c] = slot;
    }
    __syncthreads();
    const int n_dep_all = n_dep_s;
    const int n_dep = min(n_dep_all, OBS_DEP_MAX);
    // stage dependent planes: wave w takes slots w, w+4, ...; lane = env (coalesced 256-byte reads)
    for (int d = w; d < n_dep; d += OBS_THREADS / 64) {
        const float* p = obs_plane(a, dep_src_s[d]);
        if (lane < n_rows) dep_s[d][lane] = fmaf(p[env0 + lane], dep_scale_s[d], dep_base_s[d]);
    }
    __syncthreads();

    // this lane's column groups: env-independent values and dependent slots in registers
    float exo[GPL][VEC];
    int slot[GPL][VEC];
    bool any_dep[GPL];
#pragma unroll
    for (int k = 0; k < GPL; ++k) {
        any_dep[k] = false;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int c = (lane + 64 * k) * VEC + v;
            const bool live = c < seg_n;
            exo[k][v] = live ? trow[c0 + c] : 0.0f;
            slot[k][v] = live ? slot_s[c] : -1;
            any_dep[k] |= slot[k][v] >= 0;
        }
    }
    for (int e = w; e < n_rows; e += OBS_THREADS / 64) {
        float* out = a.obs + (long long)(env0 + e) * a.pitch + c0;
#pragma unroll
        for (int k = 0; k < GPL; ++k) {
            const int c = (lane + 64 * k) * VEC;
            if (c >= seg_w) continue;
            float v[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = exo[k][j];
            if (any_dep[k]) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int s = slot[k][j];
                    if (s >= 0) {
                        if (s < OBS_DEP_MAX) v[j] = dep_s[s][e];
                        else v[j] = fmaf(obs_plane(a, a.col_src[c0 + c + j])[env0 + e], a.col_scale[c0 + c + j], v[j]);
                    }
                }
            }
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(out + c) = make_float4(v[0], v[1], v[2], v[3]);
            else out[c] = v[0];
        }
    }
}

// ObsArgs + the host-side list of env-dependent columns (<= OBS_DEP_MAX), all in the kernel arguments: scalar loads, no round trip
struct ObsTileArgs {
    ObsArgs o;
    int n_deps;
    cl_obs_dep deps[OBS_DEP_MAX];
};

// Row-wise kernel with the dependent-column list in the kernel arguments (round 4; wide observation vectors in one segment -- every
// shipped schema).  cl_observe_kernel above makes THREE dependent global round trips before its first store -- col_src scan -> plane
// loads -> template-row loads, each behind a barrier -- and at 65 536 envs every workgroup of the launch is resident at once, so nobody
// stores while everybody waits: 27.4 us with the 34 dependent columns of the 2022 district against 21.8 us without them (and 19.3 vs
// 12.4 us at the 245 columns of the 2020 one), although the dependent planes are 7 % of the traffic.  Here every load of the workgroup --
// the dependent planes (wave w: columns w, w + 4, ...; lane = env), the lane's 16-byte groups of the template row, the scalar reads of
// the list's base values -- is issued before the first wait: ONE round trip, then a barrier that waits on LDS only, then stores.
// GROUPS = 16-byte column groups per lane: 2 covers 512 columns, 4 covers OBS_SEG.
// SLOTS = dependent columns one lane may own (host: falls back to cl_observe_kernel beyond 4).  The store loop holds no LDS access and
// no branch: a lane's dependent values for the 16 rows of its wave are read from the staging tile in one batch behind the barrier, the
// loop selects (v_cndmask) and stores.  (The first form read the tile inside the loop, under the lane's exec mask, column by column:
// two dependent LDS round trips and a dozen exec-mask branches per row and column group.)
template <int GROUPS, int SLOTS>
__global__ __launch_bounds__(OBS_THREADS) void cl_observe_row1_kernel(ObsTileArgs t) {
    __shared__ float dep_s[OBS_DEP_MAX][OBS_TILE + 1];      // (+ 1: lanes that read different columns of one row hit different banks)
    const ObsArgs& a = t.o;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int env0 = blockIdx.x * OBS_TILE;
    const int n_rows = min(OBS_TILE, a.n_env - env0);
    const float* __restrict__ trow = a.row + (a.env_row0 ? (long long)a.env_row0[env0 / CL_ROW0_BLOCK] * a.n_cols : 0);
    const int n_deps = a.all_exo ? 0 : t.n_deps;
    constexpr int NWV = OBS_THREADS / 64, PER_WAVE = OBS_DEP_MAX / NWV, ROWS = OBS_TILE / NWV;
    float pv[PER_WAVE];
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int d = w + k * NWV;
        pv[k] = 0.0f;
        if (d < n_deps && lane < n_rows) pv[k] = obs_plane(a, t.deps[d].src)[env0 + lane];
    }
    float exo[GROUPS][4];
    const bool row16 = (a.n_cols & 3) == 0;              // table rows start on 16-byte boundaries
#pragma unroll
    for (int k = 0; k < GROUPS; ++k) {
        const int c = (lane + 64 * k) * 4;
        if (row16 && c < a.n_cols) {
            const float4 v = *reinterpret_cast<const float4*>(trow + c);
            exo[k][0] = v.x; exo[k][1] = v.y; exo[k][2] = v.z; exo[k][3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) exo[k][j] = c + j < a.n_cols ? trow[c + j] : 0.0f;
        }
    }
    // this lane's dependent columns (scalar walk over the kernel-argument list): position k * 4 + j in its groups, list entry d
    int own_kj[SLOTS], own_d[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) { own_kj[s] = -1; own_d[s] = 0; }
    int cnt = 0;
    for (int d = 0; d < n_deps; ++d) {
        const int col = t.deps[d].col;                                   // scalar load
        const int grp = col >> 2;
        if ((grp & 63) == lane) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (cnt == s) { own_kj[s] = (grp >> 6) * 4 + (col & 3); own_d[s] = d; }
            ++cnt;
        }
    }
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int d = w + k * NWV;
        if (d < n_deps) dep_s[d][lane] = fmaf(pv[k], t.deps[d].scale, trow[t.deps[d].col]);
    }
    __syncthreads();
    float val[SLOTS][ROWS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) val[s][i] = dep_s[own_d[s]][w + i * NWV];        // (unowned slots read entry 0: discarded below)
    }
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int e = w + i * NWV;
        if (e >= n_rows) break;                                                        // wave-uniform
        float* out = a.obs + (long long)(env0 + e) * a.pitch;
#pragma unroll
        for (int k = 0; k < GROUPS; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (c >= a.padded) continue;
            float v[4] = {exo[k][0], exo[k][1], exo[k][2], exo[k][3]};
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = own_kj[s] == k * 4 + j ? val[s][i] : v[j];
            *reinterpret_cast<float4*>(out + c) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// (A persistent form -- about one workgroup per CU, the dependent planes of ALL its env tiles fetched before its first store -- was
//  built on the idea that the ~7 us the dependent columns cost are reads issued under a saturated write stream.  Measured slower, and
//  slower still with fewer waves: 65 536 x 476 34.8 us with eight waves per CU, 37.9 us with four, against 27.6 us for one tile per
//  workgroup = sixteen waves per CU (profiles/r04_observe_persistent.log).  Not kept: what the store stream wants is waves.)
// Fast path (single segment, <= OBS_DEP_MAX dependent columns -- every real schema): the R x pitch block of envs a
// workgroup writes is ONE contiguous, 256-byte aligned stretch of the output.  A template of R identical rows (the
// env-independent values) is built once in LDS; per block only the dependent columns are patched, then the buffer is
// streamed out linearly: ds_read_b128 + 16-byte global store, 1 KB contiguous per wave instruction, whole cache lines
// only.  The dependent-column list arrives in the kernel arguments (scalar loads), so the only global round trip
// before the first store is the one that fetches the template row and the planes, side by side.  Barriers wait on LDS
// traffic only (lgkmcnt), never on the outstanding stores.
// Measured (MI355X, 65 536 envs): 52 columns 11.4 us vs 14.9 us for the row-wise kernel, 476 columns 29.2 vs 27.2 us --
// the host picks this kernel for narrow observation vectors, where row-wise lanes would idle.
// (A persistent variant with a dedicated loader wave prefetching the planes of the next block was built and measured
// slower: 40 us at 476 columns -- the per-block hand-off serialises on the read latency under a saturated write stream.)
constexpr int OBS_BUF = 8192;       // floats in the LDS tile buffer

CL_DEV void obs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NT: non-temporal stores, for the small dependent-column matrix of the compact form (step + observe at 17 x 65 536: 18.0 -> 17.0 us,
// at 262 144 envs 48.8 -> 44.9 us); the full [E][n_obs] matrix gains nothing at 125 MB and loses 7 % at 500 MB.
typedef float obs_f4 __attribute__((ext_vector_type(4)));
template <bool NT>
CL_DEV void obs_st(float4* p, float4 v) {
    if constexpr (NT) { const obs_f4 x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, reinterpret_cast<obs_f4*>(p)); }
    else *p = v;
}

template <bool NT>
CL_DEV void obs_stream_out(const float* buf, float* out, int total4, int tid) {
    const float4* src4 = reinterpret_cast<const float4*>(buf);
    float4* dst4 = reinterpret_cast<float4*>(out);
    int q = tid;
    for (; q + 3 * OBS_THREADS < total4; q += 4 * OBS_THREADS) {       // 4 LDS reads in flight per lane
        const float4 v0 = src4[q], v1 = src4[q + OBS_THREADS], v2 = src4[q + 2 * OBS_THREADS], v3 = src4[q + 3 * OBS_THREADS];
        obs_st<NT>(dst4 + q, v0); obs_st<NT>(dst4 + q + OBS_THREADS, v1); obs_st<NT>(dst4 + q + 2 * OBS_THREADS, v2); obs_st<NT>(dst4 + q + 3 * OBS_THREADS, v3);
    }
    for (; q < total4; q += OBS_THREADS) obs_st<NT>(dst4 + q, src4[q]);
}

// Every column depends on the env (the compact observation form: VectorCityLearnEnv(observations='compact') hands the kernel
// only the env-dependent columns): the launch is a transpose of n_cols planes [E] into rows [E][pitch].  A workgroup owns 64
// envs: wave w reads the planes of columns w, w + 4, ... (lane = env: coalesced 256-byte reads, all issued before the first
// wait), the affine map is applied on the way into an LDS tile, and the tile -- 64 x pitch contiguous floats of `obs` -- streams
// out in 16-byte stores.  One memory round trip, no template row, no per-block patching (the narrow-vector tile kernel spent
// 14.9 us on the 17 x 65 536 x 34 case: latency-bound block rounds).
constexpr int OBS_TCOLS = 64;       // most columns the transpose kernel takes (= OBS_DEP_MAX: the list travels in the kernel arguments)
__global__ __launch_bounds__(OBS_THREADS) void cl_observe_transpose_kernel(ObsTileArgs t) {
    __shared__ __attribute__((aligned(16))) float tile[OBS_TILE * (OBS_TCOLS + 4)];
    const ObsArgs& a = t.o;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int env0 = blockIdx.x * OBS_TILE;
    const int n_rows = min(OBS_TILE, a.n_env - env0);
    const float* __restrict__ trow = a.row + (a.env_row0 ? (long long)a.env_row0[env0 / CL_ROW0_BLOCK] * a.n_cols : 0);
    const int P = a.pitch;                                   // multiple of 4 (checked on the host): rows of the tile stay 16-byte aligned
    constexpr int PER_WAVE = OBS_TCOLS / (OBS_THREADS / 64);
    float v[PER_WAVE];
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int d = w + k * (OBS_THREADS / 64);
        v[k] = 0.0f;
        if (d < t.n_deps && lane < n_rows) v[k] = obs_plane(a, t.deps[d].src)[env0 + lane];
    }
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int d = w + k * (OBS_THREADS / 64);
        if (d < t.n_deps) tile[lane * P + t.deps[d].col] = fmaf(v[k], t.deps[d].scale, trow[t.deps[d].col]);
    }
    for (int c = a.n_cols + w; c < a.padded; c += OBS_THREADS / 64) tile[lane * P + c] = 0.0f;      // pad columns
    __syncthreads();
    if (a.padded == P) obs_stream_out<true>(tile, a.obs + (long long)env0 * P, n_rows * P / 4, tid);
    else {
        // rows wider than what is written (a view into a larger buffer): row by row, 16 bytes per lane
        const int q4 = a.padded / 4;
        for (int i = tid; i < n_rows * q4; i += OBS_THREADS) {
            const int r = i / q4, c = (i - r * q4) * 4;
            *reinterpret_cast<float4*>(a.obs + (long long)(env0 + r) * P + c) = *reinterpret_cast<const float4*>(tile + r * P + c);
        }
    }
}

__global__ __launch_bounds__(OBS_THREADS) void cl_observe_tile_kernel(ObsTileArgs t) {
    __shared__ __attribute__((aligned(16))) float buf[OBS_BUF];
    __shared__ float dep_s[OBS_DEP_MAX][OBS_TILE];
    __shared__ int dep_col_s[OBS_DEP_MAX];

    // The workgroup owns OBS_TILE / R blocks of R consecutive envs, interleaved with the other workgroups
    // (block b = s * gridDim.x + blockIdx.x): the resident workgroups always write one contiguous stretch.
    const ObsArgs& a = t.o;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int W = a.pitch, N = a.n_cols, R = a.sub_rows;
    const int lr = 31 - __builtin_clz(R);               // R is a power of two
    const int n_blocks = (a.n_env + R - 1) >> lr;
    const int wg = blockIdx.x, n_wg = gridDim.x;
    const int n_deps = a.all_exo ? 0 : t.n_deps;
    {   // dependent planes: lane l <-> (block l / R, row l % R); R = 16 envs = one 64-byte line of a plane
        const int b = (lane >> lr) * n_wg + wg;
        const int env = (b << lr) + (lane & (R - 1));
        const bool live = b < n_blocks && env < a.n_env;
        for (int d = w; d < n_deps; d += OBS_THREADS / 64) {
            const cl_obs_dep dep = t.deps[d];                            // wave-uniform: scalar loads
            const float base = a.row[dep.col];
            const float* p = obs_plane(a, dep.src);
            if (live) dep_s[d][lane] = fmaf(p[env], dep.scale, base);
            if (lane == 0) dep_col_s[d] = dep.col;
        }
    }
    for (int c = tid; c < W; c += OBS_THREADS) {
        const float v = c < N ? a.row[c] : 0.0f;
        for (int r = 0; r < R; ++r) buf[r * W + c] = v;
    }
    __syncthreads();
    // no VMEM loads from here on: the stores of one block stay in flight while the next block is patched and read
    for (int sb = 0; sb < (OBS_TILE >> lr); ++sb) {
        const int b = sb * n_wg + wg;
        if (b >= n_blocks) break;
        const int env0 = b << lr;
        const int rows = min(R, a.n_env - env0);
        for (int idx = tid; idx < (n_deps << lr); idx += OBS_THREADS) {
            const int d = idx >> lr, r = idx & (R - 1);
            if (r < rows) buf[r * W + dep_col_s[d]] = dep_s[d][(sb << lr) + r];
        }
        if (n_deps) obs_lds_barrier();
        obs_stream_out<false>(buf, a.obs + (long long)env0 * W, (rows * W) >> 2, tid);   // n_env % 4 == 0 -> rows % 4 == 0
        if (n_deps) obs_lds_barrier();
    }
}

// Wave-independent kernel (wide observation vectors with a host-side dependent-column list): no LDS, no barriers, no
// atomics.  A wave owns OBS_WROWS consecutive envs; lane l owns the 16-byte column groups l, l + 64, ... whose
// env-independent values stay in registers.  A lane that owns a dependent column fetches that plane's values for the
// wave's envs itself -- OBS_WROWS x 4 B = one 64-byte line -- so every load of the kernel is issued at wave start in one
// batch (one memory round trip before the first store) and waves never wait for each other: the hardware overlaps the
// read latency of starting waves with the stores of running ones.
constexpr int OBS_WROWS = 16;       // envs per wave
constexpr int OBS_WSLOTS = 4;       // dependent columns a lane can own (host falls back to the row-wise kernel beyond)
// OBS_WGROUPS = 16-byte column groups per lane: 2 covers 512 columns (every shipped schema), 4 covers OBS_SEG
template <int OBS_WGROUPS>
__global__ __launch_bounds__(OBS_THREADS) void cl_observe_wave_kernel(ObsTileArgs t) {
    const ObsArgs& a = t.o;
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (OBS_THREADS / 64) + (threadIdx.x >> 6);
    const int env0 = gw * OBS_WROWS;
    if (env0 >= a.n_env) return;                                     // wave-uniform
    const int rows = min(OBS_WROWS, a.n_env - env0);
    const float* __restrict__ trow = a.row + (a.env_row0 ? (long long)a.env_row0[env0 / CL_ROW0_BLOCK] * a.n_cols : 0);
    const int n_deps = a.all_exo ? 0 : t.n_deps;

    float exo[OBS_WGROUPS][4];
#pragma unroll
    for (int k = 0; k < OBS_WGROUPS; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (lane + 64 * k) * 4 + j;
            exo[k][j] = c < a.n_cols ? trow[c] : 0.0f;
        }
    }
    // this lane's dependent columns (scalar walk over the kernel-argument list)
    int s_kj[OBS_WSLOTS];                       // k * 4 + j of the slot, -1 = empty
    int s_src[OBS_WSLOTS];
    float s_scale[OBS_WSLOTS];
#pragma unroll
    for (int s = 0; s < OBS_WSLOTS; ++s) { s_kj[s] = -1; s_src[s] = 0; s_scale[s] = 0.0f; }
    int cnt = 0;
    for (int d = 0; d < n_deps; ++d) {
        const cl_obs_dep dep = t.deps[d];                            // scalar loads
        const int grp = dep.col >> 2;
        if ((grp & 63) == lane) {
            const int kj = (grp >> 6) * 4 + (dep.col & 3);
#pragma unroll
            for (int s = 0; s < OBS_WSLOTS; ++s)
                if (cnt == s) { s_kj[s] = kj; s_src[s] = dep.src; s_scale[s] = dep.scale; }
            ++cnt;
        }
    }
    // fetch the dependent values: one 64-byte line per owned column (n_env % 4 == 0: whole float4s are in range)
    float val[OBS_WSLOTS][OBS_WROWS];
#pragma unroll
    for (int s = 0; s < OBS_WSLOTS; ++s) {
        if (s_kj[s] >= 0) {
            const float4* p = reinterpret_cast<const float4*>(obs_plane(a, s_src[s]) + env0);
#pragma unroll
            for (int q = 0; q < OBS_WROWS / 4; ++q) {
                const float4 v = 4 * q < rows ? p[q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                val[s][4 * q] = v.x; val[s][4 * q + 1] = v.y; val[s][4 * q + 2] = v.z; val[s][4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < OBS_WROWS; ++e) val[s][e] = 0.0f;
        }
    }
#pragma unroll
    for (int s = 0; s < OBS_WSLOTS; ++s) {
        float base = 0.0f;
#pragma unroll
        for (int k = 0; k < OBS_WGROUPS; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) base = s_kj[s] == k * 4 + j ? exo[k][j] : base;
#pragma unroll
        for (int e = 0; e < OBS_WROWS; ++e) val[s][e] = fmaf(val[s][e], s_scale[s], base);
    }
    float* out = a.obs + (long long)env0 * a.pitch;
#pragma unroll
    for (int e = 0; e < OBS_WROWS; ++e) {
#pragma unroll
        for (int k = 0; k < OBS_WGROUPS; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (e >= rows || c >= a.padded) continue;                 // `e >= rows` is wave-uniform
            float v[4] = {exo[k][0], exo[k][1], exo[k][2], exo[k][3]};
            if (cnt) {
#pragma unroll
                for (int s = 0; s < OBS_WSLOTS; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = s_kj[s] == k * 4 + j ? val[s][e] : v[j];
            }
            *reinterpret_cast<float4*>(out + (long long)e * a.pitch + c) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

}  // namespace
