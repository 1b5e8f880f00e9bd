// vbx_autocorr.hpp -- wave-cooperative autocorrelation of one LDS-resident frame.
//
// Reference: Autocorrelate::autocorrelate_mut, src/periodic.rs:276-289
//   r[lag] = x[0] + sum_{i=1}^{N-lag-1} x[i] * x[i+lag]          (fold seeded with x[0], Q1)
// With the frame zero-padded past N this equals  S[lag] - x[0]*x[lag] + x[0]  where
// S[lag] = sum_{i>=0} x[i]*x[i+lag]; the kernels compute S and apply the correction.
//
// Mapping ("lag tiles"): lags are processed in passes of 64*KT consecutive lags; in a pass
// lane l owns the KT consecutive lags  base + l*KT + k.  Stepping i by one slides each lane's
// KT-wide sample window by one element, so a step costs ONE ds_read_b64 (stride KT doubles
// across lanes: conflict-free for odd KT) and KT FMAs; x[i], x[i+1] come from one broadcast
// ds_read_b128 (all lanes read the same address).  Pass g only runs i < N - g*64*KT, which
// trims most of the triangle's empty half.
#pragma once

#include "vbx_device.hpp"

namespace vbx {

constexpr int AC_KT = 5;                    // lags per lane per pass (odd -> conflict-free window reads)
constexpr int AC_PASS = VBX_WAVE * AC_KT;   // lags per pass (320)

// number of zero doubles that must follow the N samples in LDS
__host__ __device__ constexpr int autocorr_pad(int /*n*/) { return AC_PASS + VBX_WAVE; }

// xs: LDS (16-byte aligned), samples [0,N) followed by >= autocorr_pad(N) zeros.
// Calls emit(lag, S_lag) for every lag in [0, n_lags) from the lane that owns it.
template <typename Emit>
__device__ __forceinline__ void autocorr_tiles(const double *xs, int n, int n_lags, Emit emit) {
    const int lane = lane_id();
    constexpr int UN = 2 * AC_KT;                    // steps per unrolled group (two window rotations)
    for (int base = 0; base < n_lags; base += AC_PASS) {
        const int trips = n - base;                  // i < N - base contributes to some lag of this pass
        const int lag0 = base + lane * AC_KT;
        double acc[AC_KT];
        double win[AC_KT];
#pragma unroll
        for (int k = 0; k < AC_KT; k++) { acc[k] = 0.0; win[k] = xs[lag0 + k]; }
        const double *wp = xs + lag0 + AC_KT;        // next window sample of this lane
        for (int i0 = 0; i0 < trips; i0 += UN) {     // steps past `trips` only meet the zero pad
            double2 xb[AC_KT];
#pragma unroll
            for (int q = 0; q < AC_KT; q++) xb[q] = *reinterpret_cast<const double2 *>(xs + i0 + 2 * q);
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const double xi = (u & 1) ? xb[u >> 1].y : xb[u >> 1].x;
                const double nxt = wp[i0 + u];
#pragma unroll
                for (int k = 0; k < AC_KT; k++) acc[k] = fma(xi, win[(k + u) % AC_KT], acc[k]);
                win[u % AC_KT] = nxt;                // that slot held the oldest sample
            }
        }
#pragma unroll
        for (int k = 0; k < AC_KT; k++)
            if (lag0 + k < n_lags) emit(lag0 + k, acc[k]);
    }
}

}  // namespace vbx
