// k_mfcc.hip -- MFCC::mfcc (src/spectrum.rs:401-441, Q14) and dct (:384-398).
//
// The reference takes a full complex FFT (rustfft) but only reads |X_k| of the bins below the
// last mel point (< 245 of 1200 at 48 kHz / (100, 8000) Hz), and only their magnitudes.  The
// kernel evaluates exactly those bins with the Goertzel recurrence in Reinsch's numerically
// stable form: one wavefront per frame, lane <-> bin (BPL bins per lane), the samples read
// coalesced from HBM 64 at a time and broadcast through v_readlane, no twiddle table, no LDS
// traffic in the loop.  Per bin and sample:  t = d - kappa*s;  d = sigma*t + x;  s = sigma*s + d
// (kappa = 4 sin^2(w/2), sigma = +1 when cos w > 0, else kappa = 4 cos^2(w/2), sigma = -1), and
//   |X_k|^2 = d^2 + sigma*kappa*s_{N-1}*s_{N-2}     (error grows only linearly in N).
// Mel energies are then summed per filter in the reference's order (rising |X|^2 slope, then
// the "falling" slope that also rises and uses |X|), log10 clamped from below at 1e-10, and the
// K x K DCT-II (x2) is applied from a host-built cosine table.
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

constexpr int MFCC_BPL = 4;                 // bins per lane and pass
constexpr int MFCC_PASS = 64 * MFCC_BPL;    // bins per pass over the frame

// LDS per wave: mag2[nb] | mag[nb] | en[64]
template <int W>
__global__ __launch_bounds__(64 * W) void mfcc_kernel(
    const double *__restrict__ x, long n_frames, int n, long stride, const double *__restrict__ window,
    const double *__restrict__ kappa_sigma /* [nb][2] */, const int32_t *__restrict__ bins,
    const double *__restrict__ dct_table, int num_coeffs, int nb, double *__restrict__ out) {
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const long f = (long)blockIdx.x * W + wave;
    if (f >= n_frames) return;
    const int b_lo = bins[0];
    double *mag2 = smem + (size_t)wave * (2 * (size_t)nb + 64), *mag = mag2 + nb, *en = mag + nb;
    const double *xf = x + f * stride;

    for (int p0 = 0; p0 < nb; p0 += MFCC_PASS) {
        double kap[MFCC_BPL], sig[MFCC_BPL], s[MFCC_BPL], d[MFCC_BPL];
#pragma unroll
        for (int j = 0; j < MFCC_BPL; j++) {
            const int bi = p0 + j * 64 + lane;
            const bool ok = bi < nb;
            kap[j] = ok ? kappa_sigma[2 * bi] : 0.0;
            sig[j] = ok ? kappa_sigma[2 * bi + 1] : 1.0;
            s[j] = 0.0; d[j] = 0.0;
        }
        for (int i0 = 0; i0 < n; i0 += 64) {
            double chunk = 0.0;
            if (i0 + lane < n) { chunk = xf[i0 + lane]; if (window != nullptr) chunk *= window[i0 + lane]; }
            const int steps = min(64, n - i0);
            for (int q = 0; q < steps; q++) {
                const double xi = readlane_f64(chunk, q);
#pragma unroll
                for (int j = 0; j < MFCC_BPL; j++) {
                    const double t = fma(-kap[j], s[j], d[j]);
                    d[j] = fma(sig[j], t, xi);
                    s[j] = fma(sig[j], s[j], d[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < MFCC_BPL; j++) {
            const int bi = p0 + j * 64 + lane;
            if (bi < nb) {
                const double s2 = sig[j] * (s[j] - d[j]);
                double m2 = fma(d[j], d[j], sig[j] * kap[j] * s[j] * s2);     // norm_sqr (:426)
                m2 = (m2 < 0.0) ? 0.0 : m2;
                mag2[bi] = m2;
                mag[bi] = sqrt(m2);                                           // norm (:432)
            }
        }
    }
    wave_sync();       // mag2/mag are produced and consumed inside this wavefront

    // mel energies (:421-437): lane w <-> filter w, sequential sums in the reference's order
    if (lane < num_coeffs) {
        const int w0 = bins[lane], w1 = bins[lane + 1], w2 = bins[lane + 2];
        const int up = w1 - w0, down = w2 - w1;
        double up_sum = 0.0, down_sum = 0.0;
        for (int i = 0; i < up; i++) up_sum = up_sum + fabs(mag2[w0 + i - b_lo]) * ((double)i / (double)up);
        for (int i = 0; i < down; i++) down_sum = down_sum + fabs(mag[w1 + i - b_lo]) * ((double)i / (double)down);
        const double lg = log10(up_sum + down_sum);
        en[lane] = (lg != lg || lg < 1.0e-10) ? 1.0e-10 : lg;   // f64::max(1e-10): NaN yields the other operand
    }
    wave_sync();
    if (lane < num_coeffs) {                              // dct (:391-397)
        double acc = 0.0;
        for (int j = 0; j < num_coeffs; j++) acc = acc + en[j] * dct_table[lane * num_coeffs + j];
        out[f * (long)num_coeffs + lane] = 2.0 * acc;
    }
}

// dct on rows: one lane per output coefficient, one block per row
__global__ void dct_rows_kernel(const double *__restrict__ in, long rows, int n, const double *__restrict__ dct_table,
                                double *__restrict__ out) {
    const long row = blockIdx.x;
    if (row >= rows) return;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        double acc = 0.0;
        for (int j = 0; j < n; j++) acc = acc + in[row * (long)n + j] * dct_table[(long)k * n + j];
        out[row * (long)n + k] = 2.0 * acc;
    }
}

static size_t mfcc_lds(int nb, int w) { return (size_t)w * (2 * (size_t)nb + 64) * sizeof(double); }

bool mfcc_fits(int /*n*/, int nb) { return mfcc_lds(nb, 1) <= 160 * 1024; }

void launch_mfcc(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                 const double *kappa_sigma, const int32_t *bins_dev, const double *dct_table,
                 int num_coeffs, double *out, int32_t * /*status*/, int nb) {
    if (mfcc_lds(nb, 4) <= 40 * 1024) {
        hipLaunchKernelGGL((mfcc_kernel<4>), dim3((unsigned)((F + 3) / 4)), dim3(256), mfcc_lds(nb, 4), s,
                           x, F, n, stride, window, kappa_sigma, bins_dev, dct_table, num_coeffs, nb, out);
    } else {
        hipLaunchKernelGGL((mfcc_kernel<1>), dim3((unsigned)F), dim3(64), mfcc_lds(nb, 1), s,
                           x, F, n, stride, window, kappa_sigma, bins_dev, dct_table, num_coeffs, nb, out);
    }
}

void launch_dct_rows(hipStream_t s, const double *in, long rows, int n, const double *dct_table, double *out) {
    hipLaunchKernelGGL(dct_rows_kernel, dim3((unsigned)rows), dim3(64), 0, s, in, rows, n, dct_table, out);
}

}  // namespace vbx
