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
// across lanes: conflict-free for odd KT) and KT FMAs; x[i] itself is broadcast from a
// register through v_readlane (no LDS traffic).  Pass g only runs i < N - g*64*KT, which
// trims most of the triangle's empty half.
#pragma once

#include "vbx_device.hpp"

namespace vbx {

constexpr int AC_KT = 5;                    // lags per lane per pass (odd -> conflict-free window reads)
constexpr int AC_PASS = VBX_WAVE * AC_KT;   // lags per pass (320)

// number of zero doubles that must follow the N samples in LDS
__host__ __device__ constexpr int autocorr_pad(int /*n*/) { return AC_PASS + VBX_WAVE; }

// xs: LDS, samples [0,N) followed by >= autocorr_pad(N) zeros.
// Calls emit(lag, S_lag) for every lag in [0, n_lags) from the lane that owns it.
template <typename Emit>
__device__ __forceinline__ void autocorr_tiles(const double *xs, int n, int n_lags, Emit emit) {
    const int lane = lane_id();
    for (int base = 0; base < n_lags; base += AC_PASS) {
        const int trips = n - base;                  // i < N - base contributes to some lag of this pass
        const int lag0 = base + lane * AC_KT;
        double acc[AC_KT];
        double win[AC_KT];
#pragma unroll
        for (int k = 0; k < AC_KT; k++) { acc[k] = 0.0; win[k] = xs[lag0 + k]; }
        for (int i0 = 0; i0 < trips; i0 += 60) {     // 60 = 12 * AC_KT steps per broadcast chunk
            const double xchunk = xs[i0 + lane];     // lanes 0..59 are used (reads past N hit the zero pad)
            const int steps = min(60, trips - i0);
            for (int s0 = 0; s0 < steps; s0 += AC_KT) {
                // AC_KT unrolled steps; the window rotates through the registers by renaming
#pragma unroll
                for (int u = 0; u < AC_KT; u++) {
                    const double xi = readlane_f64(xchunk, s0 + u);
                    const double nxt = xs[i0 + s0 + u + lag0 + AC_KT];
#pragma unroll
                    for (int k = 0; k < AC_KT; k++) acc[k] = fma(xi, win[(k + u) % AC_KT], acc[k]);
                    win[u] = nxt;                    // slot u is the oldest after this step
                }
            }
        }
#pragma unroll
        for (int k = 0; k < AC_KT; k++)
            if (lag0 + k < n_lags) emit(lag0 + k, acc[k]);
    }
}

}  // namespace vbx
