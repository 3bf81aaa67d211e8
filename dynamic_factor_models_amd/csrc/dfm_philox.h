// dfm_philox.h -- Philox4x32-10 counter-based generator (Salmon et al. 2011): every number is a pure function
// of (key, counter), so results do not depend on the launch geometry.  Shared by synth.hip and boot.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfm {

struct Philox {
    static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    static __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    static __device__ __forceinline__ void block(uint64_t key, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) {
        uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
        uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round(c, k0, k1);
            k0 += W0; k1 += W1;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = c[i];
    }
};

}  // namespace dfm
