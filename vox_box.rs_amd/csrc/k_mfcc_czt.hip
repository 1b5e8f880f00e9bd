// k_mfcc_czt.hip -- host side of the chirp-z MFCC kernels (vbx_mfcc_czt.hpp): which transform a frame length takes, the
// chirp tables, the dispatch.  mfcc_czt_kernel<4> (complex 4096) is instantiated in k_mfcc_czt_u4.hip (compile time).
#include "vbx_mfcc_czt.hpp"

#include <cmath>
#include <vector>

namespace vbx {

// the smallest plan whose transform holds the convolution (L >= n + top - 1), or SPECTRAL_PLAN_NONE
int mfcc_czt_plan(int n, int top) {
    const long need = (long)n + top - 1;
    if (n < 2 || top < 1 || top > n) return SPECTRAL_PLAN_NONE;
    if (need <= 1024) return SPECTRAL_PLAN_1024;
    if (need <= 2048) return SPECTRAL_PLAN_2048;
    if (need <= 4096) return SPECTRAL_PLAN_4096;
    return SPECTRAL_PLAN_NONE;
}

int mfcc_czt_split_plan(int n, int top, int *n1) {
    const int half = (n + 1) / 2;
    *n1 = half;
    if (n < 4 || top < 1 || top > half) return SPECTRAL_PLAN_NONE;
    return mfcc_czt_plan(half, top);
}

// in-place radix-2 FFT (forward, e^{-2 pi i jk / L}) in long double: the chirp's transform is computed once per (n, L)
static void fft_ld(std::vector<long double> &re, std::vector<long double> &im) {
    const size_t L = re.size();
    for (size_t i = 1, j = 0; i < L; i++) {
        size_t bit = L >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (size_t len = 2; len <= L; len <<= 1) {
        for (size_t i = 0; i < L; i += len)
            for (size_t k = 0; k < len / 2; k++) {
                const long double ang = -two_pi * (long double)k / (long double)len;
                const long double wr = cosl(ang), wi = sinl(ang);
                const size_t a = i + k, b = i + k + len / 2;
                const long double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] += tr; im[a] += ti;
            }
    }
}

// h_chirp[2 n]: conj(w_i) = e^{-i pi i^2 / n};  h_bhat[block][2 L]: FFT_L of g_b, g_b[m mod L] = w_{m - s_b} for m in (-n_b, top),
// 0 elsewhere (s_b = b n1 the block's first sample, n_b its length; one block with s_0 = 0 unless the frame is split)
void mfcc_czt_fill_tabs(int n, int top, int L, int n1, double *h_chirp, double *h_bhat) {
    const long double pi = 3.141592653589793238462643383279502884L;
    auto w = [&](long m, long double &c, long double &s) {   // w_m: the angle pi m^2 / n with m^2 reduced mod 2n exactly
        const long r = (long)(((long long)m * (long long)m) % (2LL * n));
        const long double ang = pi * (long double)r / (long double)n;
        c = cosl(ang); s = sinl(ang);
    };
    for (long i = 0; i < n; i++) { long double c, s; w(i, c, s); h_chirp[2 * i] = (double)c; h_chirp[2 * i + 1] = (double)(-s); }
    if (n1 <= 0 || n1 >= n) n1 = n;
    const int nblk = (n + n1 - 1) / n1;
    for (int b = 0; b < nblk; b++) {
        const long s_b = (long)b * n1, n_b = (n - s_b < n1) ? n - s_b : n1;
        std::vector<long double> br((size_t)L, 0.0L), bi((size_t)L, 0.0L);
        for (long m = -(n_b - 1); m < top; m++) {
            long double c, s; w(m - s_b, c, s);
            const size_t idx = (size_t)(((m % L) + L) % L);
            br[idx] = c; bi[idx] = s;
        }
        fft_ld(br, bi);
        double *o = h_bhat + (size_t)b * 2 * (size_t)L;
        for (long k = 0; k < L; k++) { o[2 * k] = (double)br[(size_t)k]; o[2 * k + 1] = (double)bi[(size_t)k]; }
    }
}

void launch_mfcc_czt_u4(hipStream_t s, const double *x, long F, int n, int n1, long stride, const double *window, const double *tab,
                        const double *chirp, const double *bhat, const int32_t *bins, const double *slopes, const double *dct,
                        int num_coeffs, int nb, double *out, long out_ld, int32_t *status, double *cw_scratch);

void launch_mfcc_czt(hipStream_t s, int plan, const double *x, long F, int n, int n1, long stride, const double *window, const double *tab,
                     const double *chirp, const double *bhat, const int32_t *bins, const double *slopes, const double *dct,
                     int num_coeffs, int nb, double *out, long out_ld, int32_t *status, double *cw_scratch) {
    if (plan == SPECTRAL_PLAN_1024) launch_mfcc_czt_u<1, 1>(s, x, F, n, n1, stride, window, tab, chirp, bhat, bins, slopes, dct, num_coeffs, nb, out, out_ld, status, cw_scratch);
    else if (plan == SPECTRAL_PLAN_2048) launch_mfcc_czt_u<2, 1>(s, x, F, n, n1, stride, window, tab, chirp, bhat, bins, slopes, dct, num_coeffs, nb, out, out_ld, status, cw_scratch);
    else launch_mfcc_czt_u4(s, x, F, n, n1, stride, window, tab, chirp, bhat, bins, slopes, dct, num_coeffs, nb, out, out_ld, status, cw_scratch);
}

}  // namespace vbx
