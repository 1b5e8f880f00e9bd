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
#include "vbx_mfcc_tail.hpp"

namespace vbx {

constexpr int MFCC_BPL = 4;                 // bins per lane and pass
constexpr int MFCC_PASS = 64 * MFCC_BPL;    // bins per pass over the frame

// mel energies (:421-437) and dct (:391-397) of one frame (one wavefront).  pu[b] = |X_b|^2 * (i / up) and
// pd[b] = |X_b| * (i / down) are the reference's per-bin products (the slope factors i/up, i/down come from a
// host table: the same IEEE division, done once); the sums run sequentially in the reference's order.
__device__ __forceinline__ void mfcc_tail(const double *pu, const double *pd, double *en, const int32_t *bins,
                                          const double *dct_table, int num_coeffs, int b_lo, int lane,
                                          double *out_row) {
    if (lane < num_coeffs) {                              // lane w <-> filter w
        const int w0 = bins[lane], w1 = bins[lane + 1], w2 = bins[lane + 2];
        // four LDS reads in flight per step; entries past a filter's end are replaced by +0.0, which leaves
        // the running sum unchanged, so the order and rounding of the reference's loop are kept
        double up_sum = 0.0, down_sum = 0.0;
        for (int b = w0 - b_lo; b < w1 - b_lo; b += 4) {
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (b + j < w1 - b_lo) ? pu[b + j] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) up_sum = up_sum + v[j];
        }
        for (int b = w1 - b_lo; b < w2 - b_lo; b += 4) {
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (b + j < w2 - b_lo) ? pd[b + j] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) down_sum = down_sum + v[j];
        }
        const double lg = log10(up_sum + down_sum);
        en[lane] = (lg != lg || lg < 1.0e-10) ? 1.0e-10 : lg;   // f64::max(1e-10): NaN yields the other operand
    }
    wave_sync();
    if (lane < num_coeffs) {                              // dct (:391-397)
        double acc = 0.0;
        for (int j = 0; j < num_coeffs; j++) acc = acc + en[j] * dct_table[lane * num_coeffs + j];
        out_row[lane] = 2.0 * acc;
    }
}

// LDS per wave: mag2[nb] | mag[nb] | en[64]
template <int W>
__global__ __launch_bounds__(64 * W) void mfcc_kernel(
    const double *__restrict__ x, long n_frames, int n, long stride, const double *__restrict__ window,
    const double *__restrict__ kappa_sigma /* [nb][2] */, const int32_t *__restrict__ bins,
    const double *__restrict__ slopes /* [nb][2] i/up, i/down */,
    const double *__restrict__ dct_table, int num_coeffs, int nb, double *__restrict__ out, long out_ld, int32_t *__restrict__ status) {
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
                const double2 sl = *reinterpret_cast<const double2 *>(slopes + 2 * bi);
                mag2[bi] = fabs(m2) * sl.x;                                   // norm_sqr * multiplier (:426-428)
                mag[bi] = fabs(sqrt(m2)) * sl.y;                              // norm * multiplier (:432-434)
            }
        }
    }
    wave_sync();       // mag2/mag are produced and consumed inside this wavefront

    mfcc_tail(mag2, mag, en, bins, dct_table, num_coeffs, b_lo, lane, out + f * out_ld);
        if (status != nullptr && lane == 0) status[f] = 0;
}

// ------------------------------------------------------------------------------------------
// Two-stage DFT of the needed bins (frame_len = n1 * n2).  With n = n2*i1 + i2 and k = k1 + n1*k2:
//   A[i2][k1] = sum_i1 x[n2*i1 + i2] * W_n1^(i1*k1)            stage 1: one real [n2 x n1]*[n1 x n1] product
//   X[k]      = sum_i2 A[i2][k mod n1] * W_n^(i2*k)             stage 2: n2 complex MACs per needed bin
// which is n*n1 + 4*nb*n2 real MACs per frame instead of Goertzel's 3*n*nb.  The input is real, so stage 1
// computes the cos columns k1 = 0..n1/2 and the sin columns k1 = 1..(n1-1)/2 only (n1 real columns, padded to
// NC); the other half follows from A[i2][n1-k1] = conj A[i2][k1].  One wavefront per frame: each lane owns a
// TM x 4 register tile of A (x broadcast from LDS, table rows as ds_read_b128), then one needed bin per pass in
// stage 2 (twiddles W_n^j from a host-built table held in LDS).  Tables are rounded from long double.
// LDS: per block the stage-1 table C[n1][NC] and the twiddles [n][2]; per wave xs[max(n, 2 nb)] (reused for |X|^2, |X|) | A[n2][NC] | en[64].
// ------------------------------------------------------------------------------------------
template <int TM>
__global__ __launch_bounds__(512) void mfcc_dft2_kernel(
    const double *__restrict__ x, long n_frames, int n, long stride, const double *__restrict__ window,
    const double *__restrict__ ctab /* [n1][NC] */, const double *__restrict__ twid /* [n][2] cos, sin */,
    int n1, int n2, int NC, int xs_len, const int32_t *__restrict__ bins, const double *__restrict__ slopes,
    const double *__restrict__ dct_table, int num_coeffs, int nb, double *__restrict__ out, long out_ld, int32_t *__restrict__ status) {
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id(), W = blockDim.x >> 6;
    double *ct = smem, *tw = smem + (size_t)n1 * NC;                      // shared by the block
    const size_t per_wave = (size_t)xs_len + (size_t)n2 * NC + 64;
    double *xs = tw + 2 * (size_t)n + (size_t)wave * per_wave, *A = xs + xs_len, *en = A + (size_t)n2 * NC;
    for (int i = threadIdx.x; i < n1 * NC; i += blockDim.x) ct[i] = ctab[i];
    for (int i = threadIdx.x; i < 2 * n; i += blockDim.x) tw[i] = twid[i];
    __syncthreads();

    const int b_lo = bins[0];
    const int H = n1 >> 1, ncos = H + 1;
    const int nct = NC >> 2;                                             // column tiles of 4
    const int nrt = (n2 + TM - 1) / TM;                                  // row tiles of TM
    const int ntiles = nrt * nct;
    // The window (the same for every frame) and the NEXT frame's first 64*PF samples live in registers: the
    // loads of frame f+1 are in flight while frame f is transformed (1-2 waves per SIMD hide nothing else).
    constexpr int PF = 20;
    double wreg[PF], pre[PF];
    const long fstep = (long)gridDim.x * W;
    long f = (long)blockIdx.x * W + wave;
#pragma unroll
    for (int j = 0; j < PF; j++) {
        const int i = 64 * j + lane;
        wreg[j] = (window != nullptr && i < n) ? window[i] : 1.0;
        pre[j] = (f < n_frames && i < n) ? x[f * stride + i] : 0.0;
    }
    for (; f < n_frames; f += fstep) {
        const double *xf = x + f * stride;
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const int i = 64 * j + lane;
            if (i < n) xs[i] = pre[j] * wreg[j];
        }
        for (int i = 64 * PF + lane; i < n; i += 64) xs[i] = (window != nullptr) ? xf[i] * window[i] : xf[i];
        if (f + fstep < n_frames) {
            const double *xn = xf + fstep * stride;
#pragma unroll
            for (int j = 0; j < PF; j++) {
                const int i = 64 * j + lane;
                pre[j] = (i < n) ? xn[i] : 0.0;
            }
        }
        wave_sync();
        // stage 1
        for (int t = lane; t < ((ntiles + 63) & ~63); t += 64) {
            const bool live = t < ntiles;
            const int rt = live ? t / nct : 0, ctile = live ? t % nct : 0;
            int roff[TM];
#pragma unroll
            for (int i = 0; i < TM; i++) { const int r = rt * TM + i; roff[i] = (r < n2) ? r : n2 - 1; }
            double acc[TM][4];
#pragma unroll
            for (int i = 0; i < TM; i++) { acc[i][0] = 0.; acc[i][1] = 0.; acc[i][2] = 0.; acc[i][3] = 0.; }
            const double *cp = ct + 4 * ctile;
            const double *xp = xs;
#pragma unroll 4
            for (int i1 = 0; i1 < n1; i1++) {
                const double2 c01 = *reinterpret_cast<const double2 *>(cp), c23 = *reinterpret_cast<const double2 *>(cp + 2);
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    const double xv = xp[roff[i]];
                    acc[i][0] = fma(xv, c01.x, acc[i][0]);
                    acc[i][1] = fma(xv, c01.y, acc[i][1]);
                    acc[i][2] = fma(xv, c23.x, acc[i][2]);
                    acc[i][3] = fma(xv, c23.y, acc[i][3]);
                }
                cp += NC; xp += n2;
            }
            if (live) {
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    const int r = rt * TM + i;
                    if (r < n2) {
                        double *ap = A + (size_t)r * NC + 4 * ctile;
                        *reinterpret_cast<double2 *>(ap) = make_double2(acc[i][0], acc[i][1]);
                        *reinterpret_cast<double2 *>(ap + 2) = make_double2(acc[i][2], acc[i][3]);
                    }
                }
            }
        }
        wave_sync();
        // stage 2: the frame's samples are dead, their space takes |X|^2 and |X|
        double *mag2 = xs, *mag = xs + nb;
        for (int p0 = 0; p0 < nb; p0 += 64 * MFCC_BPL) {                 // MFCC_BPL independent bins per lane
            int kk[MFCC_BPL], cc[MFCC_BPL], sc[MFCC_BPL], idx[MFCC_BPL];
            double sgn[MFCC_BPL], re[MFCC_BPL], im[MFCC_BPL];
#pragma unroll
            for (int j = 0; j < MFCC_BPL; j++) {
                const int bi = p0 + 64 * j + lane;
                kk[j] = b_lo + ((bi < nb) ? bi : 0);
                const int k1 = kk[j] % n1;
                cc[j] = (k1 <= H) ? k1 : n1 - k1;                        // column of Re A
                const bool has_im = cc[j] >= 1 && 2 * cc[j] != n1;
                sc[j] = has_im ? ncos + cc[j] - 1 : 0;                   // column of -Im A[.][cc]
                sgn[j] = has_im ? ((k1 <= H) ? -1.0 : 1.0) : 0.0;
                re[j] = 0.; im[j] = 0.; idx[j] = 0;
            }
            const double *ap = A;
            for (int i2 = 0; i2 < n2; i2++) {
#pragma unroll
                for (int j = 0; j < MFCC_BPL; j++) {
                    const double2 w = *reinterpret_cast<const double2 *>(tw + 2 * idx[j]);   // cos, sin of 2 pi idx / n
                    const double ar = ap[cc[j]];
                    const double as = ap[sc[j]];
                    const double ai = (sgn[j] != 0.0) ? sgn[j] * as : 0.0;
                    re[j] = fma(ar, w.x, re[j]); re[j] = fma(ai, w.y, re[j]);                // (ar + i ai)(cos - i sin)
                    im[j] = fma(ai, w.x, im[j]); im[j] = fma(-ar, w.y, im[j]);
                    idx[j] += kk[j]; idx[j] -= (idx[j] >= n) ? n : 0;
                }
                ap += NC;
            }
#pragma unroll
            for (int j = 0; j < MFCC_BPL; j++) {
                const int bi = p0 + 64 * j + lane;
                if (bi < nb) {
                    const double m2 = fma(re[j], re[j], im[j] * im[j]);
                    const double2 sl = *reinterpret_cast<const double2 *>(slopes + 2 * bi);
                    mag2[bi] = fabs(m2) * sl.x;                          // norm_sqr * multiplier (:426-428)
                    mag[bi] = fabs(sqrt(m2)) * sl.y;                     // norm * multiplier (:432-434)
                }
            }
        }
        wave_sync();
        mfcc_tail(mag2, mag, en, bins, dct_table, num_coeffs, b_lo, lane, out + f * out_ld);
        if (status != nullptr && lane == 0) status[f] = 0;
        wave_sync();                                                     // xs / en
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
                 const double *kappa_sigma, const int32_t *bins_dev, const double *slopes, const double *dct_table,
                 int num_coeffs, double *out, long out_ld, int32_t *status, int nb) {
    if (mfcc_lds(nb, 4) <= 40 * 1024) {
        hipLaunchKernelGGL((mfcc_kernel<4>), dim3((unsigned)((F + 3) / 4)), dim3(256), mfcc_lds(nb, 4), s,
                           x, F, n, stride, window, kappa_sigma, bins_dev, slopes, dct_table, num_coeffs, nb, out, out_ld, status);
    } else {
        hipLaunchKernelGGL((mfcc_kernel<1>), dim3((unsigned)F), dim3(64), mfcc_lds(nb, 1), s,
                           x, F, n, stride, window, kappa_sigma, bins_dev, slopes, dct_table, num_coeffs, nb, out, out_ld, status);
    }
}

// ---- two-stage plan: geometry shared by the host (table builder in vbx_api.hip) and the launcher ----
size_t mfcc_dft2_lds(const mfcc_plan_t &pl, int nb, int n, int waves) {
    const size_t xs_len = (size_t)((n > 2 * nb) ? n : 2 * nb);
    return ((size_t)pl.n1 * pl.nc + 2 * (size_t)n + (size_t)waves * (xs_len + (size_t)pl.n2 * pl.nc + 64)) * sizeof(double);
}

// Picks n1*n2 = n minimising n*n1 + 8*nb*n2 (stage 2 is the less efficient loop) and the row tile TM that
// fills the lanes best; ok = false when n has no useful factorisation (the Goertzel kernel then runs).
mfcc_plan_t mfcc_plan(int n, int nb) {
    mfcc_plan_t best{};
    best.ok = false;
    double best_cost = 0.5 * 3.0 * (double)n * (double)nb;     // must at least halve Goertzel's work
    for (int n1 = 2; n1 <= n / 2; n1++) {
        if (n % n1) continue;
        const int n2 = n / n1;
        if (n1 > 128 || n2 > 256) continue;
        const int nc = (n1 + 3) & ~3, nct = nc / 4;
        int tm_best = 0; double tile_cost = 0.0;
        for (int tm : {2, 3, 4, 5, 6, 8}) {
            const int tiles = ((n2 + tm - 1) / tm) * nct;
            const double c = (double)((tiles + 63) / 64) * 64.0 * (double)n1 * (4.0 * tm + 0.75 * (tm + 2));
            if (tm_best == 0 || c < tile_cost) { tm_best = tm; tile_cost = c; }
        }
        const double cost = tile_cost + 8.0 * (double)(((nb + 63) / 64) * 64) * (double)n2;
        mfcc_plan_t pl{true, n1, n2, nc, tm_best};
        if (mfcc_dft2_lds(pl, nb, n, 1) > 160 * 1024) continue;
        if (cost < best_cost) { best_cost = cost; best = pl; }
    }
    return best;
}

template <int TM>
static void launch_dft2_tm(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                           const mfcc_plan_t &pl, const double *ctab, const double *twid, const int32_t *bins_dev,
                           const double *slopes, const double *dct_table, int num_coeffs, double *out, long out_ld, int32_t *status, int nb, int cu_count) {
    // waves per block: the most wavefronts per CU the LDS allows (the table is shared by the block)
    int best_w = 1, best_res = 0;
    for (int w = 1; w <= 8; w++) {
        const size_t b = mfcc_dft2_lds(pl, nb, n, w);
        if (b > 160 * 1024) break;
        const int res = (int)((160 * 1024) / b) * w;
        if (res > best_res || (res == best_res && w <= 4)) { best_res = res; best_w = w; }
    }
    const int xs_len = (n > 2 * nb) ? n : 2 * nb;
    long blocks = (F + best_w - 1) / best_w;
    const long cap = (long)(cu_count > 0 ? cu_count : 256) * 16;       // grid-stride: the table is loaded once per block
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((mfcc_dft2_kernel<TM>), dim3((unsigned)blocks), dim3(64 * best_w), mfcc_dft2_lds(pl, nb, n, best_w), s,
                       x, F, n, stride, window, ctab, twid, pl.n1, pl.n2, pl.nc, xs_len, bins_dev, slopes, dct_table,
                       num_coeffs, nb, out, out_ld, status);
}

void launch_mfcc_dft2(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                      const mfcc_plan_t &pl, const double *ctab, const double *twid, const int32_t *bins_dev,
                      const double *slopes, const double *dct_table, int num_coeffs, double *out, long out_ld, int32_t *status, int nb, int cu_count) {
#define VBX_DFT2(TM) case TM: launch_dft2_tm<TM>(s, x, F, n, stride, window, pl, ctab, twid, bins_dev, slopes, dct_table, num_coeffs, out, out_ld, status, nb, cu_count); break;
    switch (pl.tm) { VBX_DFT2(2) VBX_DFT2(3) VBX_DFT2(4) VBX_DFT2(5) VBX_DFT2(6) VBX_DFT2(8) default: break; }
#undef VBX_DFT2
}

// rows of a constant + a constant status (frames the reference panics on before looking at the data)
__global__ void fill_rows_kernel(double *__restrict__ out, long n_rows, int n, long ld, double value,
                                 int32_t *__restrict__ status, int32_t code) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * n) return;
    const long row = i / n; const int k = (int)(i - row * n);
    out[row * ld + k] = value;
    if (k == 0 && status != nullptr) status[row] = code;
}

void launch_fill_rows(hipStream_t s, double *out, long rows, int n, long ld, double value, int32_t *status, int32_t code) {
    const long total = rows * n;
    hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out, rows, n, ld, value, status, code);
}

void launch_dct_rows(hipStream_t s, const double *in, long rows, int n, const double *dct_table, double *out) {
    hipLaunchKernelGGL(dct_rows_kernel, dim3((unsigned)rows), dim3(64), 0, s, in, rows, n, dct_table, out);
}

// The deferred tail of MFCC::mfcc (src/spectrum.rs:434-439, :391-397; vbx_mfcc_tail.hpp mfcc_tail_q with defer): every row holds its
// num_coeffs mel filter sums; log10 clamped at 1e-10 (f64::max: NaN yields the other operand), then the DCT-II x 2, in place, one row
// per lane.  The operations of mfcc_tail_q's last two steps in their order (bit-identical rows: tools/experiments/bitcompare_libs.py).
__global__ __launch_bounds__(64) void mfcc_rows_kernel(double *__restrict__ rows, long n_rows, long ld, int num_coeffs,
                                                       const double *__restrict__ dct_table) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    mfcc_row_tail(rows + row * ld, num_coeffs, dct_table);
}

void launch_mfcc_rows(hipStream_t s, double *rows, long F, long ld, int num_coeffs, const double *dct) {
    hipLaunchKernelGGL(mfcc_rows_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, s, rows, F, ld, num_coeffs, dct);
}

}  // namespace vbx
