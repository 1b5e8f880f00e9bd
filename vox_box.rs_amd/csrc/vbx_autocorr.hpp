// vbx_autocorr.hpp -- wave-cooperative autocorrelation of one LDS-resident frame.
//
// Reference: Autocorrelate::autocorrelate_mut, src/periodic.rs:276-289
//   r[lag] = x[0] + sum_{i=1}^{N-lag-1} x[i] * x[i+lag]          (fold seeded with x[0], Q1)
// With the frame zero-padded past N this equals  S[lag] - x[0]*x[lag] + x[0]  where
// S[lag] = sum_{i>=0} x[i]*x[i+lag]; the kernels compute S and apply the correction.
#pragma once

#include "vbx_device.hpp"

namespace vbx {

// ------------------------------------------------------------------------------------------
// All-lag autocorrelation on the FP64 matrix cores (v_mfma_f64_16x16x4_f64, 64 cycles, 2048 flop).
//
// One MFMA tile holds 256 consecutive lags: with  A[row][k] = z[a + k - 16 row]  and  B[k][col] = z[a + k + col + L0]
//   D[row][col] = sum_a z[a - 16 row] * z[a - 16 row + (L0 + 16 row + col)] = S[L0 + 16 row + col]
// when a runs over [0, n - L0) (z is the frame, zero outside [0, n)).  Every product the sum needs appears in
// exactly one tile, there is nothing to reduce afterwards, and accumulator register r of lane l is lag
// L0 + 64 r + l: the result is stored coalesced.  A is the same for every tile, so one K-step costs one gather
// read plus one read per live tile (lags L0 >= n - a have run out of products and are skipped).
//
// LDS image: logical index i in [-AC_MF_FRONT, n + AC_MF_BACK) lives at phys(i) = j + (j >> 4), j = i + AC_MF_FRONT
// (one pad double per 16): the A gather walks rows 16 samples apart, 17 doubles apart in LDS -> all 64 banks.
// ------------------------------------------------------------------------------------------
typedef double vbx_d4 __attribute__((ext_vector_type(4)));

constexpr int AC_MF_FRONT = 256;            // zeros before the frame (A reaches back 15*16 samples)
constexpr int AC_MF_BACK = 80;              // zeros after it (B reaches 3 + 15 samples past the last product, +4 read ahead)
constexpr int AC_MF_TILE = 256;             // lags per tile
constexpr int AC_MF_NT = 5;                 // tiles per pass (5 accumulators = 40 VGPRs): 1280 lags

__host__ __device__ constexpr int ac_mf_phys(int i) { return (i + AC_MF_FRONT) + ((i + AC_MF_FRONT) >> 4); }
__host__ __device__ constexpr int ac_mf_lds_doubles(int n) { return ac_mf_phys(n + AC_MF_BACK) + 1; }

constexpr int AC_MF_TILE_PHYS = AC_MF_TILE + AC_MF_TILE / 16;   // a tile further = 272 doubles further in the image

// Chunks of 16 samples (4 K-steps) a0, a0+16, ... < a1 with L live tiles.  pa[u] / pb[u] are the lane's operand
// addresses of K-step u of the current chunk; a chunk further is 17 doubles further (the pad double per 16 makes
// the mapping irregular INSIDE a chunk only, which the four per-lane pointers absorb): no index arithmetic in
// the loop, tiles are immediate offsets.  Returns the first a not processed.
template <int L>
__device__ __forceinline__ int ac_mf_segment(const double *(&pa)[4], const double *(&pb)[4], int a0, int a1,
                                             vbx_d4 (&acc)[AC_MF_NT]) {
    int a = a0;
    if (a >= a1) return a;
    // operands of K-step u+1 are read before the MFMAs of K-step u are issued (the image is padded for the one
    // step read past the end)
    double av = *pa[0], bv[L];
#pragma unroll
    for (int t = 0; t < L; t++) bv[t] = pb[0][t * AC_MF_TILE_PHYS];
    for (; a < a1; a += 16) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int un = (u + 1) & 3;
            if (u == 3) {
#pragma unroll
                for (int q = 0; q < 4; q++) { pa[q] += 17; pb[q] += 17; }
            }
            const double av_n = *pa[un];
            double bv_n[L];
#pragma unroll
            for (int t = 0; t < L; t++) bv_n[t] = pb[un][t * AC_MF_TILE_PHYS];
#pragma unroll
            for (int t = 0; t < L; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[t], acc[t], 0, 0, 0);
            av = av_n;
#pragma unroll
            for (int t = 0; t < L; t++) bv[t] = bv_n[t];
        }
    }
    return a;
}

// zs: LDS image described above.  Calls emit(slot, lag, S_lag) for the AC_MF_NT*4 accumulator registers of every
// pass: slot = 4 t + r is a compile-time index (results can be kept in a register array), lag = l0 + 256 t + 64 r +
// lane may be >= n_lags (the caller guards).
template <typename Emit>
__device__ __forceinline__ void autocorr_mfma(const double *zs, int n, int n_lags, Emit emit) {
    const int lane = lane_id();
    const int row = lane & 15, k = lane >> 4;
    for (int l0 = 0; l0 < n_lags; l0 += AC_MF_NT * AC_MF_TILE) {
        const double *pa[4], *pb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            pa[u] = zs + ac_mf_phys(4 * u + k - 16 * row);             // A: z[a + k - 16 row]
            pb[u] = zs + ac_mf_phys(4 * u + k + row + l0);             // B: z[a + k + col + L0]   (col = lane & 15)
        }
        vbx_d4 acc[AC_MF_NT];
#pragma unroll
        for (int t = 0; t < AC_MF_NT; t++) acc[t] = vbx_d4{0.0, 0.0, 0.0, 0.0};
        const int a_end = n - l0;                              // tile t has products for a < a_end - 256 t
        const int want_raw = (n_lags - l0 + AC_MF_TILE - 1) / AC_MF_TILE;        // tiles that hold requested lags
        const int want = (want_raw < AC_MF_NT) ? want_raw : AC_MF_NT;
        int a = 0;
        while (a < a_end) {       // a tile past its last product only meets zeros, so `live` may be taken per chunk
            const int have = (a_end - a + AC_MF_TILE - 1) / AC_MF_TILE;          // tiles that still have products (uniform)
            const int live = (have < want) ? have : want;
            const int seg_end = a_end - (live - 1) * AC_MF_TILE;                  // live stays the same for a < seg_end
            if (live == 5) a = ac_mf_segment<5>(pa, pb, a, seg_end, acc);
            else if (live == 4) a = ac_mf_segment<4>(pa, pb, a, seg_end, acc);
            else if (live == 3) a = ac_mf_segment<3>(pa, pb, a, seg_end, acc);
            else if (live == 2) a = ac_mf_segment<2>(pa, pb, a, seg_end, acc);
            else a = ac_mf_segment<1>(pa, pb, a, seg_end, acc);
        }
#pragma unroll
        for (int t = 0; t < AC_MF_NT; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                emit(4 * t + r, l0 + t * AC_MF_TILE + 64 * r + lane, acc[t][r]);
            }
        }
    }
}

}  // namespace vbx
