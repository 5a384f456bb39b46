// cl_trace.h -- wave-timeline stamps for scripts/wave_timeline.py.  Everything here compiles to nothing unless the library is
// built with -DCL_TRACE (the diagnostic libcitylearn_amd_trace.so; the product build carries none of it).
#pragma once
// Wave-timeline instrumentation, compiled only into the diagnostic library scripts/wave_timeline.py builds (-DCL_TRACE; the product
// build carries none of it): REFCLK (100 MHz) stamps of a wave's phases, parked in LDS and written out by lane 0 at the end.
#ifdef CL_TRACE
namespace { __device__ unsigned long long* g_cl_trace = nullptr; }
#define CL_TRACE_SLOTS 16
#define CL_TRACE_DECL __shared__ unsigned long long tr_lds[16][CL_TRACE_SLOTS]; \
    if (lane < CL_TRACE_SLOTS) tr_lds[w][lane] = 0ull
// The stamps inside the building loop are NOT `volatile` and carry no "memory" clobber: either one makes the asm a possible writer of
// all memory, every later parameter read stops being provably invariant and the scalar loads come back as vector loads (59 instead
// of 14 global loads, 3.5 x the launch time).  They order themselves against the code through register operands instead: a value
// passes through the asm ("+v"), so its producers come before the stamp and its consumers after.
#define CL_TRACE_ENTRY(k) do { unsigned long long t_; \
    asm("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "s"(a.n_env)); \
    if (lane == 0) tr_lds[w][k] = t_; } while (0)
#define CL_TRACE_CYCLES_ENTRY(k) do { unsigned long long t_; \
    asm("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "s"(a.n_bldg)); \
    if (lane == 0) tr_lds[w][k] = t_; } while (0)
// the building's inputs have all returned (its loads are the only vector-memory reads in flight)
#define CL_TRACE_INPUTS(k, in) do { unsigned long long t_; \
    asm("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" \
        : "=s"(t_), "+v"(in.soc), "+v"(in.eff), "+v"(in.deg), "+v"(in.cs), "+v"(in.hs), "+v"(in.ds), "+v"(in.a_cs), "+v"(in.a_hs), "+v"(in.a_ds), \
          "+v"(in.a_es), "+v"(in.a_cd), "+v"(in.a_hd)); \
    if (lane == 0) tr_lds[w][k] = t_; } while (0)
// every vector-memory access issued so far has returned; `v` (a loaded value) pins the stamp behind the loads' issue
#define CL_TRACE_WAITV(k, v) do { unsigned long long t_; \
    asm("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(v)); \
    if (lane == 0) tr_lds[w][k] = t_; } while (0)
// `v` has been computed (no wait on memory)
#define CL_TRACE_AFTER(k, v) do { unsigned long long t_; \
    asm("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(v)); \
    if (lane == 0) tr_lds[w][k] = t_; } while (0)
// end of the kernel: everything issued has been acknowledged
#define CL_TRACE_FLUSH() do { unsigned long long t_, c_; \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(c_) :: "memory"); \
    if (lane == 0) { tr_lds[w][CL_TRACE_SLOTS - 2] = t_; tr_lds[w][12] = c_; \
        tr_lds[w][CL_TRACE_SLOTS - 1] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); } \
    if (lane < CL_TRACE_SLOTS && g_cl_trace) \
        g_cl_trace[(((long long)blockIdx.y * gridDim.x + blockIdx.x) * 16 + w) * CL_TRACE_SLOTS + lane] = tr_lds[w][lane]; } while (0)
#else
#define CL_TRACE_DECL
#define CL_TRACE_ENTRY(k)
#define CL_TRACE_CYCLES_ENTRY(k)
#define CL_TRACE_INPUTS(k, in)
#define CL_TRACE_WAITV(k, v)
#define CL_TRACE_AFTER(k, v)
#define CL_TRACE_FLUSH()
#endif
