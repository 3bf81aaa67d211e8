// dfm_device.h -- shared device helpers for the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfm {

constexpr int kWave = 64;

__host__ __device__ constexpr int npack(int r) { return r * (r + 1) / 2; }
__host__ __device__ constexpr int pow2_ge(int r) { return r <= 1 ? 1 : r <= 2 ? 2 : r <= 4 ? 4 : r <= 8 ? 8 : r <= 16 ? 16 : 32; }

// ---------------------------------------------------------------------------------------------
// Wave-wide "transpose-reduce": every lane holds CNT (<= 64) partial sums v[0..CNT); after run()
// v[0] of lane l holds the wave-wide total of value index reduce_index<CNT>(l).  Each butterfly
// stage halves the number of values a lane carries (lanes with the stage bit clear keep the even
// value of a pair and receive the partner's copy of it, lanes with the bit set keep the odd one),
// so the whole reduction costs ~CNT shuffles + adds instead of 6*CNT.
template <int C, int OFF>
struct ReduceStage {
    static __device__ __forceinline__ void run(double* v, int lane) {
        if constexpr (OFF >= 1) {
            const bool up = (lane & OFF) != 0;
            constexpr int H = C / 2;
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const double a = v[2 * j], b = v[2 * j + 1];
                const double keep = up ? b : a;
                const double send = up ? a : b;
                v[j] = keep + __shfl_xor(send, OFF, kWave);
            }
            if constexpr ((C & 1) != 0) {
                const double a = v[C - 1];
                v[H] = a + __shfl_xor(a, OFF, kWave);
            }
            ReduceStage<(C + 1) / 2, OFF / 2>::run(v, lane);
        }
    }
};

template <int CNT>
__device__ __forceinline__ void wave_transpose_reduce(double* v, int lane) {
    static_assert(CNT >= 1 && CNT <= 64, "at most one value per lane after six halvings");
    ReduceStage<CNT, 32>::run(v, lane);
}

// Which original value index ends up in v[0] of `lane`, and whether this lane is the canonical
// (lowest) holder of it (several lanes hold the same total when CNT < 64).
template <int CNT>
__device__ __forceinline__ int reduce_index(int lane, bool& canonical) {
    int cnt[7];
    cnt[0] = CNT;
#pragma unroll
    for (int s = 0; s < 6; ++s) cnt[s + 1] = (cnt[s] + 1) / 2;
    int p = 0;
    canonical = true;
#pragma unroll
    for (int s = 5; s >= 0; --s) {
        const int off = 32 >> s;
        const int c = cnt[s], h = c / 2;
        const bool bit = (lane & off) != 0;
        if (p < h) {
            p = 2 * p + (bit ? 1 : 0);
        } else {
            p = c - 1;
            canonical = canonical && !bit;
        }
    }
    return p;
}

}  // namespace dfm
