// k_mfcc.hip -- MFCC::mfcc (src/spectrum.rs:401-441, Q14) and dct (:384-398).
//
// The reference takes a full complex FFT (rustfft) but only reads the bins below the last mel
// point (< 245 of 1200 at 48 kHz / (100, 8000) Hz).  The kernel evaluates exactly those DFT
// bins: one wavefront per frame, one lane per bin, the frame broadcast through v_readlane and
// the twiddles e^{-2 pi i k n / N} read from an LDS table (exact index k*n mod N, so there is
// no recurrence error).  Mel energies are then summed per filter in the reference's order
// (rising |X|^2 slope, then the "falling" slope that also rises and uses |X|), log10 clamped
// from below at 1e-10, and the K x K DCT-II (x2) is applied from a host-built cosine table.
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

// LDS layout per block (W waves): twiddle[n] (double2) | per wave: xs[n] | mag2[nb] | mag[nb] | en[64]
template <int W>
__global__ __launch_bounds__(64 * W) void mfcc_kernel(
    const double *__restrict__ x, long n_frames, int n, long stride, const double *__restrict__ window,
    const double *__restrict__ twiddle, const int32_t *__restrict__ bins, const double *__restrict__ dct_table,
    int num_coeffs, double *__restrict__ out) {
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int b_lo = bins[0], b_hi = bins[num_coeffs + 1];
    const int nb = b_hi - b_lo;
    double *tw = smem;                                   // [n][2]
    double *wbase = smem + 2 * (size_t)n + (size_t)wave * ((size_t)n + 2 * (size_t)nb + 64);
    double *xs = wbase, *mag2 = wbase + n, *mag = mag2 + nb, *en = mag + nb;
    for (int i = threadIdx.x; i < 2 * n; i += 64 * W) tw[i] = twiddle[i];
    const long f = (long)blockIdx.x * W + wave;
    const bool active = f < n_frames;
    if (active) {
        const double *xf = x + f * stride;
        for (int i = lane; i < n; i += 64) {
            double v = xf[i];
            if (window != nullptr) v *= window[i];
            xs[i] = v;
        }
    }
    __syncthreads();
    if (!active) return;

    // DFT of bins [b_lo, b_hi): lane <-> bin
    for (int g = 0; g < nb; g += 64) {
        const int k = b_lo + g + lane;
        const int kk = (k < b_hi) ? (k % n) : 0;
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const double chunk = (i0 + lane < n) ? xs[i0 + lane] : 0.0;
            const int steps = min(64, n - i0);
            for (int s = 0; s < steps; s++) {
                const double xi = readlane_f64(chunk, s);
                const double2 cs = *reinterpret_cast<const double2 *>(tw + 2 * idx);
                re = fma(xi, cs.x, re);
                im = fma(-xi, cs.y, im);
                idx += kk; idx = (idx >= n) ? idx - n : idx;
            }
        }
        if (k < b_hi) {
            const double m2 = re * re + im * im;          // norm_sqr (:426)
            mag2[k - b_lo] = m2;
            mag[k - b_lo] = hypot(re, im);                // norm (:432)
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

static size_t mfcc_lds(int n, int nb, int w) {
    return ((size_t)2 * n + (size_t)w * ((size_t)n + 2 * (size_t)nb + 64)) * sizeof(double);
}

// returns false if the shape does not fit the LDS
bool mfcc_fits(int n, int nb) { return mfcc_lds(n, nb, 1) <= 160 * 1024; }

void launch_mfcc(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                 const double *twiddle, const int32_t *bins_dev, const double *dct_table,
                 int num_coeffs, double *out, int32_t * /*status*/, int nb) {
    // pick the widest block whose LDS still allows >= 2 blocks per CU
    if (mfcc_lds(n, nb, 4) <= 80 * 1024) {
        hipLaunchKernelGGL((mfcc_kernel<4>), dim3((unsigned)((F + 3) / 4)), dim3(256), mfcc_lds(n, nb, 4), s,
                           x, F, n, stride, window, twiddle, bins_dev, dct_table, num_coeffs, out);
    } else if (mfcc_lds(n, nb, 2) <= 80 * 1024) {
        hipLaunchKernelGGL((mfcc_kernel<2>), dim3((unsigned)((F + 1) / 2)), dim3(128), mfcc_lds(n, nb, 2), s,
                           x, F, n, stride, window, twiddle, bins_dev, dct_table, num_coeffs, out);
    } else {
        hipLaunchKernelGGL((mfcc_kernel<1>), dim3((unsigned)F), dim3(64), mfcc_lds(n, nb, 1), s,
                           x, F, n, stride, window, twiddle, bins_dev, dct_table, num_coeffs, out);
    }
}

void launch_dct_rows(hipStream_t s, const double *in, long rows, int n, const double *dct_table, double *out) {
    hipLaunchKernelGGL(dct_rows_kernel, dim3((unsigned)rows), dim3(64), 0, s, in, rows, n, dct_table, out);
}

}  // namespace vbx
