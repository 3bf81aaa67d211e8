// collapse_wide.hip -- balanced-panel collapse and Gram matrix for cross-sections whose weights do not fit a
// wave's register file (BASELINE config 4: N = 1000, r = 20 -> 32 padded factors: Lam / R is 256 KB per
// replicate).  Same contract as collapse_mfma.hip (b_t = sum_i lam_i x_it / R_i, sum_t s_t; no missing cell),
// same instruction (`v_mfma_f64_4x4x4_4b_f64`, lane layout in collapse_mfma.hip), different blocking:
//
//   work unit = one wave x one tile of 16 periods (4 MFMA row blocks) x ALL series.  Per step of CS series the
//   wave loads the 4 A operands (one per row block) straight from the panel -- a 128-byte line serves 4
//   consecutive steps out of the vector L1 -- and the B operand W = lam / R from the L2-resident parameters,
//   reused by the 4 row blocks.  W is re-read once per tile: 8 N Rp bytes against 16 x 8 N bytes of panel, so
//   L2 carries ~3x the HBM traffic (Rp = 32) while HBM sees every panel byte once.
//   No LDS, no barrier; waves are independent.  sum_t s_t goes to scol[b][tile] (one partial per tile).
// gram_wide_kernel: C = Lam' R^-1 Lam and sum log R for the same shapes, one workgroup per replicate.
// Reference counterpart: forming Lambda' x_t in the per-period regression of x_t on Lambda
// (dfm_functions.ipynb:271-286 called from :364).
#include "dfm_gram.h"
#include "dfm_kernels.h"

namespace dfm {

template <int R>
struct WideGeo {
    static constexpr int NFG = (R + 3) / 4;
    static constexpr int FPI = NFG < 4 ? NFG : 4;
    static constexpr int NCG = 4 / FPI;
    static constexpr int NINST = NFG / FPI;
    static constexpr int CS = 4 * NCG;
};
constexpr int kWideRows = 16;    // periods per tile

template <int R>
__global__ __launch_bounds__(256) void collapse_wide_kernel(CollapseArgs a, int ntile) {
    using G = WideGeo<R>;
    constexpr int CS = G::CS, NRB = kWideRows / 4;
    const int lane = threadIdx.x & 63;
    const int unit = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= a.B * ntile) return;
    const int b = unit / ntile, tile = unit % ntile;
    const int N = a.N, T = a.T;
    const int t0 = tile * kWideRows;
    const int K = lane >> 4, blk = (lane >> 2) & 3, q = lane & 3;
    const int g = blk / G::FPI, h = blk % G::FPI;
    const double* __restrict__ X = a.panel + (size_t)b * T * N;
    const double* __restrict__ L = a.Lam + (size_t)b * N * R;
    const double* __restrict__ Rv = a.Rv + (size_t)b * N;
    const double* xrow[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        int t = t0 + 4 * rb + q;
        t = t < T ? t : T - 1;                               // clamped rows are computed and never stored
        xrow[rb] = X + (size_t)t * N;
    }
    double D[NRB][G::NINST];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int m = 0; m < G::NINST; ++m) D[rb][m] = 0.0;
    double qs = 0.0;
    const int steps = (N + CS - 1) / CS;
#pragma unroll 2
    for (int s = 0; s < steps; ++s) {
        const int c = s * CS + 4 * g + K;
        const bool own = c < N;
        const int cc = own ? c : N - 1;
        const double ri = own ? 1.0 / Rv[cc] : 0.0;
        double w[G::NINST];
#pragma unroll
        for (int m = 0; m < G::NINST; ++m) {
            const int f = 4 * (h + G::FPI * m) + q;
            w[m] = (f < R) ? L[(size_t)cc * R + (f < R ? f : R - 1)] * ri : 0.0;
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const double x = xrow[rb][cc];
            const bool valid = (t0 + 4 * rb + q) < T;
            qs = fma(valid ? x * ri : 0.0, x, qs);
#pragma unroll
            for (int m = 0; m < G::NINST; ++m) D[rb][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, w[m], D[rb][m], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int m = 0; m < G::NINST; ++m) {
            if constexpr (G::NCG >= 2) D[rb][m] += xor_lane<8>(D[rb][m]);
            if constexpr (G::NCG == 4) D[rb][m] += xor_lane<4>(D[rb][m]);
            const int t = t0 + 4 * rb + K;                   // D row = lane / 16
            const int f = 4 * (h + G::FPI * m) + q;
            if (g == 0 && t < T && f < R) a.bcol[((size_t)b * T + t) * R + f] = D[rb][m];
        }
    // every x was counted once per factor-group lane (FPI duplicates): exact power-of-two rescale
    qs = wave_allsum(qs) * (1.0 / G::FPI);
    if (lane == 0) {
        a.scol[(size_t)b * T + tile] = qs;
        if (qs != qs) atomicOr(a.status, 1);   // NaN in the panel on the balanced path
    }
}

template <int R>
__global__ __launch_bounds__(256) void gram_wide_kernel(CollapseArgs a) {
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = a.N;
    const double* __restrict__ L = a.Lam + (size_t)b * N * R;
    const double* __restrict__ Rv = a.Rv + (size_t)b * N;
    constexpr int NE = (R * R + 255) / 256;                  // entries of C per thread
    double acc[NE];
    int ei[NE], ej[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + 256 * e;
        acc[e] = 0.0;
        ei[e] = (idx / R) % R; ej[e] = idx % R;
    }
    for (int c = 0; c < N; ++c) {
        const double ri = 1.0 / Rv[c];
#pragma unroll
        for (int e = 0; e < NE; ++e) acc[e] = fma(L[(size_t)c * R + ei[e]] * ri, L[(size_t)c * R + ej[e]], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + 256 * e;
        if (idx < R * R) a.Cfull[(size_t)b * R * R + idx] = acc[e];
    }
    double ld = 0.0;
    for (int c = tid; c < N; c += 256) ld += log(Rv[c]);
    ld = wave_allsum(ld);
    if ((tid & 63) == 0) red[tid >> 6] = ld;
    __syncthreads();
    if (tid == 0) a.ldfull[b] = red[0] + red[1] + red[2] + red[3];
}

int collapse_wide_tiles(int T) { return (T + kWideRows - 1) / kWideRows; }
bool collapse_wide_supported(int Rpad, int N) { return Rpad >= 2 && Rpad <= 32 && N >= 1; }

hipError_t launch_collapse_wide(int Rpad, const CollapseArgs& a, hipStream_t s) {
    note_kernel("collapse_wide_kernel");
    const int ntile = collapse_wide_tiles(a.T);
    const long long units = (long long)a.B * ntile;
    const unsigned grid = (unsigned)((units + 3) / 4);
    switch (Rpad) {
        case 2: hipLaunchKernelGGL((collapse_wide_kernel<2>), dim3(grid), dim3(256), 0, s, a, ntile); break;
        case 4: hipLaunchKernelGGL((collapse_wide_kernel<4>), dim3(grid), dim3(256), 0, s, a, ntile); break;
        case 8: hipLaunchKernelGGL((collapse_wide_kernel<8>), dim3(grid), dim3(256), 0, s, a, ntile); break;
        case 16: hipLaunchKernelGGL((collapse_wide_kernel<16>), dim3(grid), dim3(256), 0, s, a, ntile); break;
        case 32: hipLaunchKernelGGL((collapse_wide_kernel<32>), dim3(grid), dim3(256), 0, s, a, ntile); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_gram_wide(int Rpad, const CollapseArgs& a, hipStream_t s) {
    note_kernel("gram_wide_kernel");
    switch (Rpad) {
        case 2: hipLaunchKernelGGL((gram_wide_kernel<2>), dim3(a.B), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((gram_wide_kernel<4>), dim3(a.B), dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL((gram_wide_kernel<8>), dim3(a.B), dim3(256), 0, s, a); break;
        case 16: hipLaunchKernelGGL((gram_wide_kernel<16>), dim3(a.B), dim3(256), 0, s, a); break;
        case 32: hipLaunchKernelGGL((gram_wide_kernel<32>), dim3(a.B), dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace dfm
