// dfm_device.h -- shared device helpers for the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfm {

constexpr int kWave = 64;

// "hipFuncSetAttribute(MaxDynamicSharedMemorySize) done" per DEVICE: the attribute belongs to the function on the current
// device, and the multi-GPU library object (multi.hip) launches the same kernels from one process on several devices, from
// one host thread per GPU.  Drop-in for the `static bool` flag the launchers used (`if (!flag) { ...; flag = true; }`).
struct LdsOptIn {
    unsigned long long mask = 0;
    static unsigned long long bit() { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
    bool operator!() const { return (__atomic_load_n(&mask, __ATOMIC_ACQUIRE) & bit()) == 0; }
    LdsOptIn& operator=(bool done) { if (done) __atomic_fetch_or(&mask, bit(), __ATOMIC_RELEASE); return *this; }
};

__host__ __device__ constexpr int npack(int r) { return r * (r + 1) / 2; }
__host__ __device__ constexpr int pow2_ge(int r) { return r <= 1 ? 1 : r <= 2 ? 2 : r <= 4 ? 4 : r <= 8 ? 8 : r <= 16 ? 16 : 32; }

// ---------------------------------------------------------------------------------------------
// Wave-wide "transpose-reduce": every lane holds CNT (<= 64) partial sums v[0..CNT); after run()
// v[0] of lane l holds the wave-wide total of value index reduce_index<CNT>(l).  Each butterfly
// stage halves the number of values a lane carries (lanes with the stage bit clear keep the even
// value of a pair and receive the partner's copy of it, lanes with the bit set keep the odd one),
// so the whole reduction costs ~CNT shuffles + adds instead of 6*CNT.
// ---- cross-lane moves without LDS traffic (gfx950: DPP row ops + v_permlane{16,32}_swap) --------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    // every source lane of these row permutations is valid, so `old` is a don't-care: the mov form
    // (bound_ctrl) lets the compiler write a fresh destination without first copying the source
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppXor3 = 0x1B;         // quad_perm [3,2,1,0]
constexpr int kDppHalfMirror = 0x141;  // lane i <- lane i ^ 7  (within 8)
constexpr int kDppRowMirror = 0x140;   // lane i <- lane i ^ 15 (within 16)

// value held by lane (lane ^ S), 0 <= S < 64
template <int S>
__device__ __forceinline__ double xor_lane(double v) {
    static_assert(S >= 0 && S < 64, "xor distance");
    if constexpr (S == 0) return v;
    else if constexpr (S == 1) return dpp_mov<kDppXor1>(v);
    else if constexpr (S == 2) return dpp_mov<kDppXor2>(v);
    else if constexpr (S == 3) return dpp_mov<kDppXor3>(v);
    else if constexpr (S < 8) return xor_lane<S ^ 7>(dpp_mov<kDppHalfMirror>(v));
    else if constexpr (S < 16) return xor_lane<S ^ 15>(dpp_mov<kDppRowMirror>(v));
    else if constexpr (S < 32) {   // ds_swizzle bit-mask mode: and 0x1f, or 0, xor 0x10 (no LDS memory touched)
        int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F);
        int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);
        return xor_lane<S ^ 16>(__hiloint2double(hi, lo));
    } else {
        return xor_lane<S ^ 32>(__shfl_xor(v, 32, kWave));
    }
}

// a' = {a.lo32lanes, b.lo32lanes}, b' = {a.hi32lanes, b.hi32lanes}: after the swap a' + b' is, in lanes
// 0..31, a[l] + a[l+32] and, in lanes 32..63, b[l-32] + b[l] -- one butterfly stage of the
// transpose-reduce with no select and no LDS crossbar.
__device__ __forceinline__ void swap_halves32(double& a, double& b) {
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}
// same between 16-lane rows: odd rows of a <-> even rows of b
__device__ __forceinline__ void swap_rows16(double& a, double& b) {
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}

template <int C, int OFF>
struct ReduceStage {
    static __device__ __forceinline__ void run(double* v, int lane) {
        if constexpr (OFF >= 1) {
            constexpr int H = C / 2;
            if constexpr (OFF >= 16) {
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    double a = v[2 * j], b = v[2 * j + 1];
                    if constexpr (OFF == 32) swap_halves32(a, b); else swap_rows16(a, b);
                    v[j] = a + b;
                }
                if constexpr ((C & 1) != 0) {
                    double a = v[C - 1], b = a;
                    if constexpr (OFF == 32) swap_halves32(a, b); else swap_rows16(a, b);
                    v[H] = a + b;
                }
            } else {
                const bool up = (lane & OFF) != 0;
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    const double a = v[2 * j], b = v[2 * j + 1];
                    const double keep = up ? b : a;
                    const double send = up ? a : b;
                    v[j] = keep + xor_lane<OFF>(send);
                }
                if constexpr ((C & 1) != 0) {
                    const double a = v[C - 1];
                    v[H] = a + xor_lane<OFF>(a);
                }
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
