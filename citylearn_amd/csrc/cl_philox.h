// cl_philox.h -- Philox4x32-10 counter-based random numbers shared by the rollout policy (cl_rollout.h) and the
// unconnected-EV SoC drift (cl_flex.h).  Host-callable too (cl_philox_uniform).
#pragma once
#include <stdint.h>

namespace cl {

// Philox4x32-10 (Salmon et al., SC'11).  One block yields four 32-bit words; the policy stream is defined as
//   u(seed; env, column, t) = word[t & 3] of Philox(counter = (env, column, t >> 2, 0), key = (seed lo, seed hi))
// so a unit needs one block per four time steps.
struct U4 { uint32_t w[4]; };

#ifdef __HIPCC__
__host__ __device__
#endif
inline U4 philox_block(unsigned long long seed, uint32_t env, uint32_t col, uint32_t tq) {
    uint32_t c0 = env, c1 = col, c2 = tq, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{{c0, c1, c2, c3}};
}

#ifdef __HIPCC__
__host__ __device__
#endif
inline float u01(uint32_t word) { return (float)(word >> 8) * (1.0f / 16777216.0f); }   // 24 random bits -> [0, 1)

#ifdef __HIPCC__
__host__ __device__
#endif
inline uint32_t philox_word(const U4& b, uint32_t sel) {
    // two-level select, not `b.w[sel]`: a dynamically indexed array goes through scratch memory on the GPU (a 16-byte store and a load
    // per draw inside the rollout loop)
    const uint32_t w0 = b.w[0], w1 = b.w[1], w2 = b.w[2], w3 = b.w[3];
    const uint32_t lo = (sel & 1u) ? w1 : w0, hi = (sel & 1u) ? w3 : w2;
    return (sel & 2u) ? hi : lo;
}

#ifdef __HIPCC__
__host__ __device__
#endif
inline float philox_u01(unsigned long long seed, uint32_t env, uint32_t col, uint32_t t) {
    return u01(philox_word(philox_block(seed, env, col, t >> 2), t & 3u));
}

}  // namespace cl
