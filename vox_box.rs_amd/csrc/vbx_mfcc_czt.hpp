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

// cw[b][j] = window[i] * conj(w_i) for i = b n1 + j inside block b (j < n1, i < n), 0 up to the transform's length L: the caller's
// window folded into the chirp and the zero padding made data, once per call -- the frame kernel's loads are then unconditional
// (the sample index clamped, the product with a zero entry is the padding).  One block (n1 >= n) unless the frame is split.
static __global__ void czt_fold_kernel(const double *__restrict__ window, const double2 *__restrict__ chirp, int n, int L, int n1, int nblk,
                                       double2 *__restrict__ cw) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L * nblk) return;
    const int b = t / L, j = t - b * L;
    const int i = b * n1 + j;
    double2 o = double2{0.0, 0.0};
    if (j < n1 && i < n) { const double w = (window != nullptr) ? window[i] : 1.0; const double2 c = chirp[i]; o = double2{w * c.x, w * c.y}; }
    cw[t] = o;
}

// U units per thread, W wavefronts per frame (vbx_spectral_pow2.hpp): L = 1024 <1, 1>, 2048 <2, 1>, 4096 <2, 2>
// SPLIT: the frame in blocks (below); the one-block form is its own instantiation, without the loop (registers)
template <int U, int W, bool SPLIT>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mfcc_czt_kernel(const double *x, long n_frames, int n, int n1, int nblk, long stride,
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

    // A frame longer than the transform allows in one piece (n + top - 1 > L) is SPLIT: with i = s_b + j in block b,
    //     sum_i a_i w_{k-i} = sum_b sum_j a_{s_b+j} w_{(k-j) - s_b},
    // one convolution per block with its own chirp segment g_b[m] = w_{m - s_b}, m in (-n_b, top) (bhat[b], FFT_L of it), all of
    // length L >= n_b + top - 1; the blocks' COMPLEX results add up before the magnitude is taken.  Each thread owns the same
    // bins in every pass, so the running sum of the needed ones (b_lo <= k < b_lo + nb) waits in LDS behind the exchange
    // buffer, touched by its owner alone.
    double xr[TQ][R], xi[TQ][R];
    for (int blk = 0; blk < (SPLIT ? nblk : 1); blk++) {
        const int s_b = blk * n1;
        const int n_b = (n - s_b < n1) ? n - s_b : n1;
        const double2 *cwb = chirp + (long)blk * NC;
        const double2 *bh = bhat + (long)blk * NC;
        // (the twiddle table's address is opaque in every pass: as loop-invariant loads its entries would be hoisted out of the
        // block loop and held in registers across both transforms -- 100 registers spilled against none)
        const double2 *tabb = tab;
        if constexpr (SPLIT) asm volatile("" : "+s"(tabb));
        // a_j = x_{s_b + j} * cw_j in the transform's stage-1 layout (lane l, unit u, slot q holds index 16R q + l + 64 u); cw = the
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
                    v[q] = xf[s_b + ((i < n_b) ? i : n_b - 1)];
                    c[q] = cwb[i];
                }
#pragma unroll
                for (int q = 0; q < 8; q++) { re[u][8 * h + q] = v[q] * c[q].x; im[u][8 * h + q] = v[q] * c[q].y; }
            }
        fft_pow2<U, W>(re, im, xr, xi, ex, tabb);

        // times FFT_L(chirp), conjugated for the inverse, back to the stage-1 layout through the exchange buffer: the real parts,
        // then the imaginary parts (the table is read again rather than kept: registers)
#pragma unroll
        for (int t = 0; t < TQ; t++) {
#pragma unroll
            for (int kc = 0; kc < R; kc++) {
                const int k = tid + NT * t + 256 * kc;
                const double2 b = bh[k];
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
                const double2 b = bh[k];
                ex[k] = -fma(xr[t][kc], b.y, xi[t][kc] * b.x);
            }
        }
        pow2_sync<W>();
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < 16; q++) im[u][q] = ex[16 * R * q + tid + NT * u];
        pow2_sync<W>();
        fft_pow2<U, W>(re, im, xr, xi, ex, tabb);                // = conj(L * conv): |conv[k]|^2 = (xr^2 + xi^2) / L^2
        if constexpr (SPLIT) {
            const int b_lo = bins[0];
            double *accr = smem + G::EX, *acci = accr + ((nb + 1) & ~1);
#pragma unroll
            for (int t = 0; t < TQ; t++)
#pragma unroll
                for (int kc = 0; kc < R; kc++) {
                    const int b1 = tid + NT * t + 256 * kc - b_lo;
                    if (b1 >= 0 && b1 < nb) {
                        if (blk > 0) { xr[t][kc] += accr[b1]; xi[t][kc] += acci[b1]; }
                        if (blk + 1 < nblk) { accr[b1] = xr[t][kc]; acci[b1] = xi[t][kc]; }
                    }
                }
        }
    }

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
void launch_mfcc_czt_u(hipStream_t s, const double *x, long F, int n, int n1, long stride, const double *window, const double *tab,
                       const double *chirp, const double *bhat, const int32_t *bins, const double *slopes, const double *dct,
                       int num_coeffs, int nb, double *out, long out_ld, int32_t *status, double *cw_scratch) {
    const int nblk = (n1 > 0 && n1 < n) ? (n + n1 - 1) / n1 : 1;
    if (nblk == 1) n1 = n;
    size_t lds = (size_t)pow2_geom<U, W>::EX * sizeof(double);
    const size_t mel = (size_t)(2 * ((nb + 1) & ~1) + 64) * sizeof(double);
    if (mel > lds) lds = mel;
    if (nblk > 1) lds = (size_t)pow2_geom<U, W>::EX * sizeof(double) + (mel > 2 * (size_t)((nb + 1) & ~1) * sizeof(double) ? mel : 2 * (size_t)((nb + 1) & ~1) * sizeof(double));
    constexpr int L = pow2_geom<U, W>::NC;           // cw_scratch: nblk * L double2 the caller owns
    hipLaunchKernelGGL(czt_fold_kernel, dim3((unsigned)((L * nblk + 255) / 256)), dim3(256), 0, s, window, reinterpret_cast<const double2 *>(chirp), n, L,
                       n1, nblk, reinterpret_cast<double2 *>(cw_scratch));
    chirp = cw_scratch;
    if (nblk > 1)
        hipLaunchKernelGGL((mfcc_czt_kernel<U, W, true>), dim3((unsigned)F), dim3(64 * W), lds, s, x, F, n, n1, nblk, stride,
                           reinterpret_cast<const double2 *>(tab), reinterpret_cast<const double2 *>(chirp),
                           reinterpret_cast<const double2 *>(bhat), bins, slopes, dct, num_coeffs, nb, out, out_ld, status);
    else
        hipLaunchKernelGGL((mfcc_czt_kernel<U, W, false>), dim3((unsigned)F), dim3(64 * W), lds, s, x, F, n, n1, nblk, stride,
                           reinterpret_cast<const double2 *>(tab), reinterpret_cast<const double2 *>(chirp),
                           reinterpret_cast<const double2 *>(bhat), bins, slopes, dct, num_coeffs, nb, out, out_ld, status);
}

}  // namespace vbx
