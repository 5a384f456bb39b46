// cl_tune.hip -- microbenchmarks and layout probes used by scripts/ and one layout test.  Built into
// libcitylearn_amd_tune.so, which the product library and the package never load: nothing here is on the step path.
//   cl_tune_mfma_bf16_probe : operand / accumulator layout of v_mfma_f32_32x32x16_bf16 (tests/test_gpu_lstm.py)
//   cl_tune_mfma_bench      : how MFMA chains and transcendental VALU work share a SIMD (scripts/mfma_bench.py)
//   cl_tune_copy_floor      : streaming floor of the headline step (scripts/copy_floor.py)
//   cl_tune_launch_gap      : where a back-to-back launch period goes -- waves alive vs the gap between kernels, by store policy
//                             (scripts/launch_gap.py)
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- layout probe for v_mfma_f32_32x32x16_bf16 (tests/test_gpu_lstm.py::test_bf16_mfma_operand_layout) ----
typedef __bf16 cl_bf16x8 __attribute__((ext_vector_type(8)));
typedef float cl_f32x16 __attribute__((ext_vector_type(16)));
__global__ void cl_mfma_bf16_probe_kernel(const uint16_t* A, const uint16_t* B, float* D) {
    const int l = threadIdx.x, i = l & 31, kh = l >> 5;
    union { cl_bf16x8 v; uint16_t u[8]; } a, b;
    for (int j = 0; j < 8; ++j) { a.u[j] = A[i * 16 + 8 * kh + j]; b.u[j] = B[(8 * kh + j) * 32 + i]; }
    cl_f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = c[r];
}

// ---- issue-rate microbenchmark (scripts/mfma_bench.py): how MFMA chains and transcendental VALU work share a SIMD ----
template <int MODE>
__global__ void __launch_bounds__(256) cl_mfma_bench_kernel(float* out, int iters) {
    const int l = threadIdx.x & 63;
    union { cl_bf16x8 v; uint16_t u[8]; } a, b;
    for (int j = 0; j < 8; ++j) { a.u[j] = 0x3c00 + l + j; b.u[j] = 0x3b80 + l * 3 + j; }
    const float af = 1.0f + l * 1e-3f, bfv = 0.5f + l * 1e-3f;
    cl_f32x16 c0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float x0 = l * 1e-3f, x1 = x0 + 0.1f, x2 = x0 + 0.2f, x3 = x0 + 0.3f;
#define BF(C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, C, 0, 0, 0);
#define F32(C) C = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bfv, C, 0, 0, 0);
#define VX x0 = __expf(-x0); x1 = __expf(-x1); x2 = __expf(-x2); x3 = __expf(-x3);
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) { BF(c0) BF(c0) BF(c0) BF(c0) BF(c0) BF(c0) BF(c0) BF(c0) }
        if constexpr (MODE == 1) { BF(c0) BF(c1) BF(c0) BF(c1) BF(c0) BF(c1) BF(c0) BF(c1) }
        if constexpr (MODE == 2) { BF(c0) BF(c1) BF(c2) BF(c3) BF(c0) BF(c1) BF(c2) BF(c3) }
        if constexpr (MODE == 3) { BF(c0) VX BF(c1) VX BF(c0) VX BF(c1) VX BF(c0) VX BF(c1) VX BF(c0) VX BF(c1) VX }
        if constexpr (MODE == 4) { VX VX VX VX VX VX VX VX }
        if constexpr (MODE == 5) { F32(c0) F32(c1) F32(c0) F32(c1) F32(c0) F32(c1) F32(c0) F32(c1) }
        if constexpr (MODE == 6) { F32(c0) VX F32(c1) VX F32(c0) VX F32(c1) VX F32(c0) VX F32(c1) VX F32(c0) VX F32(c1) VX }
        if constexpr (MODE == 7) { BF(c0) BF(c1) BF(c0) BF(c1) BF(c0) BF(c1) BF(c0) BF(c1) VX VX VX VX VX VX VX VX }   // blocks, not interleaved
        __builtin_amdgcn_sched_barrier(0);
    }
#undef BF
#undef F32
#undef VX
    float r = x0 + x1 + x2 + x3;
    for (int k = 0; k < 16; ++k) r += c0[k] + c1[k] + c2[k] + c3[k];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

// ---- streaming floor of the headline step (scripts/copy_floor.py): same launch shape and byte counts, no arithmetic ----
// 17 buildings x 65 536 envs: per (building, 256-env tile) read 3 state planes + 1 action plane, write 3 state planes + net +
// reward; 16 waves per workgroup, 16-byte accesses -- what cl_step_kernel<4, lean> moves, with the energy model replaced by adds.
__global__ void __launch_bounds__(1024) cl_copy_floor_kernel(const float* __restrict__ st_in, const float* __restrict__ act,
                                                            float* __restrict__ st_out, float* __restrict__ out2, int n_bldg, int n_env) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int env0 = blockIdx.x * 256 + lane * 4;
    const long long plane = (long long)n_bldg * n_env;
    for (int b = w; b < n_bldg; b += nw) {
        const long long off = (long long)b * n_env + env0;
        const float4 s0 = *reinterpret_cast<const float4*>(st_in + 0 * plane + off);
        const float4 s1 = *reinterpret_cast<const float4*>(st_in + 1 * plane + off);
        const float4 s2 = *reinterpret_cast<const float4*>(st_in + 2 * plane + off);
        const float4 a = *reinterpret_cast<const float4*>(act + off);
        const float4 x = make_float4(s0.x + a.x, s0.y + a.y, s0.z + a.z, s0.w + a.w);
        *reinterpret_cast<float4*>(st_out + 0 * plane + off) = x;
        *reinterpret_cast<float4*>(st_out + 1 * plane + off) = s1;
        *reinterpret_cast<float4*>(st_out + 2 * plane + off) = s2;
        *reinterpret_cast<float4*>(out2 + 0 * plane + off) = make_float4(s1.x + a.x, s1.y + a.y, s1.z + a.z, s1.w + a.w);
        *reinterpret_cast<float4*>(out2 + 1 * plane + off) = make_float4(s2.x + a.x, s2.y + a.y, s2.z + a.z, s2.w + a.w);
    }
}

// ---- launch-gap probe (scripts/launch_gap.py): the copy-floor access pattern with a choice of store policy, each wave stamping
// REFCLK (100 MHz) when it enters and after its last store was acknowledged.  period - (last ack - first entry) = what the command
// processor and the end-of-kernel cache maintenance cost per launch.
//   ST 0: plain stores (write-back L2, dirty lines written back by the end-of-kernel release)   1: sc1 (write-through at agent scope)
//      2: sc0 sc1 (system scope)   3: nt (streaming hint)
typedef float cl_f32x4 __attribute__((ext_vector_type(4)));
template <int ST>
__device__ __forceinline__ void cl_gap_store(float* p, float4 v4) {
    const cl_f32x4 v = {v4.x, v4.y, v4.z, v4.w};
    if constexpr (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if constexpr (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if constexpr (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if constexpr (ST == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if constexpr (ST == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
    if constexpr (ST == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    if constexpr (ST == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    if constexpr (ST == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}

template <bool NT>
__device__ __forceinline__ float4 cl_gap_load(const float* p) {
    if constexpr (NT) {
        const cl_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const cl_f32x4*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else return *reinterpret_cast<const float4*>(p);
}

template <int ST, bool NTL>
__global__ void __launch_bounds__(1024) cl_launch_gap_kernel(const float* __restrict__ st_in, const float* __restrict__ act,
                                                            float* __restrict__ st_out, float* __restrict__ out2, int n_bldg, int n_env,
                                                            int n_planes_out, unsigned long long* stamps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned long long t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    const int env0 = blockIdx.x * 256 + lane * 4;
    const long long plane = (long long)n_bldg * n_env;
    for (int b = w; b < n_bldg; b += nw) {
        const long long off = (long long)b * n_env + env0;
        const float4 s0 = cl_gap_load<NTL>(st_in + 0 * plane + off);
        const float4 s1 = cl_gap_load<NTL>(st_in + 1 * plane + off);
        const float4 s2 = cl_gap_load<NTL>(st_in + 2 * plane + off);
        const float4 a = cl_gap_load<NTL>(act + off);
        const float4 x = make_float4(s0.x + a.x, s0.y + a.y, s0.z + a.z, s0.w + a.w);
        if (n_planes_out > 0) cl_gap_store<ST>(st_out + 0 * plane + off, x);
        if (n_planes_out > 1) cl_gap_store<ST>(st_out + 1 * plane + off, s1);
        if (n_planes_out > 2) cl_gap_store<ST>(st_out + 2 * plane + off, s2);
        if (n_planes_out > 3) cl_gap_store<ST>(out2 + 0 * plane + off, make_float4(s1.x + a.x, s1.y + a.y, s1.z + a.z, s1.w + a.w));
        if (n_planes_out > 4) cl_gap_store<ST>(out2 + 1 * plane + off, make_float4(s2.x + a.x, s2.y + a.y, s2.z + a.z, s2.w + a.w));
        if (n_planes_out <= 0 && x.x == 12345.678f) st_out[off] = x.y;          // keep the loads alive in the read-only variant
    }
    unsigned long long t1;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (stamps && lane == 0) {
        const long long i = ((long long)blockIdx.x * nw + w) * 2;
        stamps[i] = t0; stamps[i + 1] = t1;
    }
}

extern "C" {

int cl_tune_launch_gap(int store_mode, const float* st_in, const float* act, float* st_out, float* out2, int n_bldg, int n_env,
                       int n_planes_out, int threads, unsigned long long* stamps, void* stream) {
    const dim3 grid(n_env / 256), block(threads);
    switch (store_mode) {
#define CL_CASE(M) case M: hipLaunchKernelGGL((cl_launch_gap_kernel<M, false>), grid, block, 0, (hipStream_t)stream, st_in, act, st_out, out2, n_bldg, n_env, n_planes_out, stamps); break; \
    case M + 8: hipLaunchKernelGGL((cl_launch_gap_kernel<M, true>), grid, block, 0, (hipStream_t)stream, st_in, act, st_out, out2, n_bldg, n_env, n_planes_out, stamps); break;
    CL_CASE(0) CL_CASE(1) CL_CASE(2) CL_CASE(3) CL_CASE(4) CL_CASE(5) CL_CASE(6) CL_CASE(7)
#undef CL_CASE
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int cl_tune_copy_floor(const float* st_in, const float* act, float* st_out, float* out2, int n_bldg, int n_env, void* stream) {
    hipLaunchKernelGGL(cl_copy_floor_kernel, dim3(n_env / 256), dim3(1024), 0, (hipStream_t)stream, st_in, act, st_out, out2, n_bldg, n_env);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int cl_tune_mfma_bench(int mode, int waves_per_simd, int iters, float* out, void* stream) {
    const dim3 grid(256 * waves_per_simd), block(256);
    switch (mode) {
#define CL_CASE(M) case M: hipLaunchKernelGGL(cl_mfma_bench_kernel<M>, grid, block, 0, (hipStream_t)stream, out, iters); break;
    CL_CASE(0) CL_CASE(1) CL_CASE(2) CL_CASE(3) CL_CASE(4) CL_CASE(5) CL_CASE(6) CL_CASE(7)
#undef CL_CASE
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int cl_tune_mfma_bf16_probe(const uint16_t* A, const uint16_t* B, float* D, void* stream) {
    hipLaunchKernelGGL(cl_mfma_bf16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // extern "C"
