// vbx_mfcc_czt.hpp -- MFCC::mfcc (src/spectrum.rs:401-441) for frame lengths that fit no transform of their own: the needed
// bins of the n-point DFT by Bluestein's chirp-z identity on the power-of-two FFT of vbx_spectral_pow2.hpp.
//
// The mel filters read X_n[k], k in [b_lo, top), of the n-point DFT -- a frequency grid (k / n) that a zero-padded
// power-of-two transform does not contain unless n divides it (the fused kernels: 512, 1024, 2048 ...; 600, 800, 1200).  For
// every other n (1102 / 1103 = 25 ms at 44.1 kHz, 1600, 3000, primes ...):
//     n k = (n^2 + k^2 - (k - n)^2) / 2   =>   X[k] = conj(w_k) * sum_i (x_i conj(w_i)) w_{k-i},     w_m = e^{i pi m^2 / n}
// a LINEAR convolution of a_i = x_i conj(w_i) (n terms) with the chirp w (needed at m in (-n, top)), i.e. a circular one of
// any length L >= n + top - 1: FFT_L(a), times the precomputed FFT_L of the chirp, inverse FFT_L -- two complex transforms of
// L = 1024, 2048 or 4096 per frame, O(L log L), against n * (top - b_lo) multiply-adds of a bin-by-bin evaluation (the Goertzel
// and two-stage kernels of k_mfcc*.hip: 1 Mflop per 1103-sample frame).  The filters only use |X[k]|^2 and |X[k]|, and
// |conj(w_k)| = 1: the convolution's own magnitude is the bin's.  The inverse transform is the forward one on the conjugate.
// Tables (host, long double, rounded once): conj(w_i) for i < n with i^2 reduced mod 2n exactly; FFT_L of the chirp.
// One wavefront per frame, layouts and exchange buffer of fft_pow2<U>.
#pragma once

#include "vbx_spectral_pow2.hpp"

namespace vbx {

// cw[i] = window[i] * conj(w_i) for i < n, 0 up to the transform's length L: the caller's window folded into the chirp and the
// zero padding made data, once per call -- the frame kernel's loads are then unconditional (the sample index clamped, the
// product with a zero entry is the padding)
static __global__ void czt_fold_kernel(const double *__restrict__ window, const double2 *__restrict__ chirp, int n, int L, double2 *__restrict__ cw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    double2 o = double2{0.0, 0.0};
    if (i < n) { const double w = (window != nullptr) ? window[i] : 1.0; const double2 c = chirp[i]; o = double2{w * c.x, w * c.y}; }
    cw[i] = o;
}

// U units per thread, W wavefronts per frame (vbx_spectral_pow2.hpp): L = 1024 <1, 1>, 2048 <2, 1>, 4096 <2, 2>
template <int U, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mfcc_czt_kernel(const double *x, long n_frames, int n, long stride,
                     // no __restrict__ on the tables: as invariant loads every twiddle of both transforms is hoisted to the top
                     // of the kernel and the transforms' own values spill (measured: 675 registers spilled against 16)
                     const double2 *tab, const double2 *chirp, const double2 *bhat,
                     const int32_t *__restrict__ bins, const double *__restrict__ slopes, const double *__restrict__ dct,
                     int num_coeffs, int nb, double *__restrict__ out, long out_ld, int32_t *__restrict__ status) {
    using G = pow2_geom<U, W>;
    constexpr int R = G::R, NC = G::NC, NT = G::NT, TQ = G::TQ;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long f = xcd_item(blockIdx.x, n_frames);
    if (f >= n_frames) return;
    const int lane = lane_id();
    const int tid = pow2_tid<W>();
    const int wave = W == 1 ? 0 : (int)(threadIdx.x >> 6);
    double *ex = smem;
    const double *xf = x + f * stride;

    // a_i = x_i * cw_i in the transform's stage-1 layout (lane l, unit u, slot q holds index 16R q + l + 64 u); cw = the
    // caller's window times conj(w), folded once per call (czt_fold_kernel): three doubles in flight per sample, not four
    double re[U][16], im[U][16];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double v[8]; double2 c[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = 16 * R * (8 * h + q) + tid + NT * u;
                v[q] = xf[(i < n) ? i : n - 1];
                c[q] = chirp[i];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) { re[u][8 * h + q] = v[q] * c[q].x; im[u][8 * h + q] = v[q] * c[q].y; }
        }
    double xr[TQ][R], xi[TQ][R];
    fft_pow2<U, W>(re, im, xr, xi, ex, tab);

    // times FFT_L(chirp), conjugated for the inverse, back to the stage-1 layout through the exchange buffer: the real parts,
    // then the imaginary parts (the table is read again rather than kept: registers)
#pragma unroll
    for (int t = 0; t < TQ; t++) {
#pragma unroll
        for (int kc = 0; kc < R; kc++) {
            const int k = tid + NT * t + 256 * kc;
            const double2 b = bhat[k];
            ex[k] = fma(xr[t][kc], b.x, -(xi[t][kc] * b.y));
        }
    }
    pow2_sync<W>();
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int q = 0; q < 16; q++) re[u][q] = ex[16 * R * q + tid + NT * u];
    pow2_sync<W>();
#pragma unroll
    for (int t = 0; t < TQ; t++) {
#pragma unroll
        for (int kc = 0; kc < R; kc++) {
            const int k = tid + NT * t + 256 * kc;
            const double2 b = bhat[k];
            ex[k] = -fma(xr[t][kc], b.y, xi[t][kc] * b.x);
        }
    }
    pow2_sync<W>();
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int q = 0; q < 16; q++) im[u][q] = ex[16 * R * q + tid + NT * u];
    pow2_sync<W>();
    fft_pow2<U, W>(re, im, xr, xi, ex, tab);                 // = conj(L * conv): |conv[k]|^2 = (xr^2 + xi^2) / L^2

    const int b_lo = bins[0];
    const int nbp = (nb + 1) & ~1;
    double *pu = ex, *pd = ex + nbp, *en = ex + 2 * nbp;
    constexpr double INV_L2 = 1.0 / ((double)NC * (double)NC);
    pow2_sync<W>();
#pragma unroll
    for (int t = 0; t < TQ; t++)
#pragma unroll
        for (int kc = 0; kc < R; kc++) {
            const int b1 = tid + NT * t + 256 * kc - b_lo;
            if (b1 >= 0 && b1 < nb) {
                const double p = fma(xr[t][kc], xr[t][kc], xi[t][kc] * xi[t][kc]) * INV_L2;
                const double2 sl = *reinterpret_cast<const double2 *>(slopes + 2 * b1);
                pu[b1] = fabs(p) * sl.x;                     // norm_sqr * multiplier (src/spectrum.rs:426-428)
                pd[b1] = fabs(sqrt(p)) * sl.y;               // norm * multiplier (:432-434)
            }
        }
    pow2_sync<W>();
    if (wave != 0) return;
    if (num_coeffs <= 16) mfcc_tail_q(pu, pd, en, bins, dct, num_coeffs, b_lo, lane, out + f * out_ld);
    else mfcc_tail_m(pu, pd, en, bins, dct, num_coeffs, b_lo, lane, out + f * out_ld);
    if (status != nullptr && lane == 0) status[f] = 0;
}

template <int U, int W>
void launch_mfcc_czt_u(hipStream_t s, const double *x, long F, int n, long stride, const double *window, const double *tab,
                       const double *chirp, const double *bhat, const int32_t *bins, const double *slopes, const double *dct,
                       int num_coeffs, int nb, double *out, long out_ld, int32_t *status, double *cw_scratch) {
    size_t lds = (size_t)pow2_geom<U, W>::EX * sizeof(double);
    const size_t mel = (size_t)(2 * ((nb + 1) & ~1) + 64) * sizeof(double);
    if (mel > lds) lds = mel;
    constexpr int L = pow2_geom<U, W>::NC;           // cw_scratch: L double2 the caller owns
    hipLaunchKernelGGL(czt_fold_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, window, reinterpret_cast<const double2 *>(chirp), n, L,
                       reinterpret_cast<double2 *>(cw_scratch));
    chirp = cw_scratch;
    hipLaunchKernelGGL((mfcc_czt_kernel<U, W>), dim3((unsigned)F), dim3(64 * W), lds, s, x, F, n, stride,
                       reinterpret_cast<const double2 *>(tab), reinterpret_cast<const double2 *>(chirp),
                       reinterpret_cast<const double2 *>(bhat), bins, slopes, dct, num_coeffs, nb, out, out_ld, status);
}

}  // namespace vbx
