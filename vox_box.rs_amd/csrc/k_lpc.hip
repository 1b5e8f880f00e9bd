// k_lpc.hip -- autocorrelation (few lags / all lags), normalize, Levinson-Durbin, and the
// fused autocorrelate -> [normalize] -> lpc kernel (BASELINE config 2: HBM-bound).
//
// Reference: src/periodic.rs:276-289 (autocorrelate_mut), src/waves.rs:60-76 (normalize),
//            src/spectrum.rs:63-92 (lpc_mut / lpc).
#include "vbx_autocorr.hpp"
#include "vbx_kernels.hpp"
#include "vbx_mfcc_tail.hpp"

namespace vbx {

// ------------------------------------------------------------------------------------------
// Few-lag path: EPL contiguous samples per lane held in registers, neighbour lanes' samples
// fetched with DPP wave shifts.  HBM-bound: 8 B/sample in, (2 * n_lags) * 8 B out.
// ------------------------------------------------------------------------------------------

// Levinson-Durbin on r[0..p] held in registers (src/spectrum.rs:63-84); ac has P+1 entries.
template <int P>
__device__ __forceinline__ void levinson_uniform(const double (&r)[P + 1], double (&ac)[P + 1]) {
    double tmp[P + 1];
    double err = r[0];
    ac[0] = 1.0;
#pragma unroll
    for (int i = 1; i <= P; i++) ac[i] = 0.0;
#pragma unroll
    for (int i = 1; i <= P; i++) {
        double acc = r[i];
#pragma unroll
        for (int j = 1; j < i; j++) acc = acc + ac[j] * r[i - j];
        const double k = -acc / err;
        ac[i] = k;
#pragma unroll
        for (int j = 0; j < P; j++) tmp[j] = ac[j];
#pragma unroll
        for (int j = 1; j < i; j++) ac[j] = ac[j] + k * tmp[i - j];
        err = err * (1.0 - k * k);
    }
}

// the conditioning probe's perturbations (levinson_rows_kernel_t below has the reasoning): sign pattern `pat` of +-d on the lag sums
constexpr int LPC_PROBE_NPAT = 4;
__device__ __forceinline__ double lpc_probe_lag(double v, double d, unsigned h, int k) { return v + ((((h >> (k & 31)) & 1u) != 0) ? d : -d); }
__host__ __device__ constexpr unsigned lpc_probe_hash(int pat) { return (unsigned)pat * 0x9E3779B1u ^ ((unsigned)pat << 7); }

#ifndef VBX_AC_FPW
#define VBX_AC_FPW 16
#endif
constexpr int AC_FPW = VBX_AC_FPW;   // frames per wavefront: short enough that the last round of waves is a small tail

// EPL: samples per lane (frame_len <= 64*EPL).  NL: number of lags computed (n_lags <= NL).
//
// One wavefront works through FPW consecutive frames:
//   per frame   coalesced 16-B loads of the lane's EPL samples (next frame prefetched), neighbour
//               samples by DPP wave shifts, NL lag products per lane in registers (EPL*NL FMAs),
//               then ONE transposing reduction through LDS for all NL sums together (a lane's NL
//               partials are written as a row, 4 lanes per lag each add 16 of the 64 partials,
//               a quad butterfly finishes) instead of NL separate 64-lane reductions;
//   per FPW frames  lane f runs Levinson-Durbin for frame f (one division chain per FPW frames, not
//               per frame) and the [FPW, NL] result blocks leave through LDS as coalesced stores.
// Two frames are kept in flight ahead of the one being reduced.
// T: the Sample type of the frames and of the outputs (double, or float for the f32 instantiation of the traits: the
// samples are widened on load, the windowed product is rounded to T first -- the reference's Windower hands the traits
// frames of T -- all sums run in f64 and the results are rounded to T once, on the store).
template <int EPL, int NL, typename T>
__global__ __launch_bounds__(64) void autocorr_fewlags_kernel(
    const T *__restrict__ x, long n_frames, int n, long stride, const T *__restrict__ window,
    int n_lags, int normalize, T *__restrict__ out_r, T *__restrict__ out_lpc, long lpc_ld,
    int32_t *__restrict__ lpc_list, int32_t *__restrict__ lpc_count) {
    constexpr int FPW = AC_FPW;
    constexpr int TS = NL | 1;                       // odd row stride: conflict-free column reads
    __shared__ double TR[64 * TS];                   // per-frame transpose buffer [lane][lag]
    __shared__ double R[FPW * TS];                   // results of the wave's frames [frame][lag]
    const int lane = lane_id();
    const long f0 = (long)blockIdx.x * FPW;
    if (f0 >= n_frames) return;
    const int nf = (int)((n_frames - f0 < FPW) ? (n_frames - f0) : FPW);

    double wreg[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        const int i = lane * EPL + e;
        wreg[e] = (i < n) ? ((window != nullptr) ? (double)window[i] : 1.0) : 0.0;
    }
    const bool full = (n == 64 * EPL);               // uniform: unguarded, mergeable loads
    // 16-byte loads where every row is 16-byte aligned: a lane's EPL samples are contiguous, and with one double per
    // instruction the 64 lanes touch 64 cache lines EPL times over
    const bool vec16 = full && sizeof(T) == 8 && EPL % 2 == 0 && (((uintptr_t)x) & 15) == 0 && (stride & 1) == 0;
    auto load_frame = [&](int g, double (&dst)[EPL]) {
        const T *xf = x + (f0 + g) * stride + lane * EPL;
        if (vec16) {
            const double2 *xv = reinterpret_cast<const double2 *>(xf);
#pragma unroll
            // (non-temporal loads here: 1.166 against 0.834 ms per million frames -- measured round 3, not adopted)
            for (int e = 0; e < EPL; e += 2) { const double2 v = xv[e / 2]; dst[e] = v.x; if (e + 1 < EPL) dst[e + 1] = v.y; }
        } else if (full) {
#pragma unroll
            for (int e = 0; e < EPL; e++) dst[e] = (double)xf[e];
        } else {
#pragma unroll
            for (int e = 0; e < EPL; e++) dst[e] = (lane * EPL + e < n) ? (double)xf[e] : 0.0;
        }
    };
    double cur[EPL], nxt[EPL], nx2[EPL];
    load_frame(0, cur);
    if (nf > 1) load_frame(1, nxt);
    const int red_lag = lane >> 2, red_part = lane & 3;

    for (int g = 0; g < nf; g++) {
        if (g + 2 < nf) load_frame(g + 2, nx2);
        // ext[0..EPL) own samples, ext[EPL..EPL+NL-1) the following samples (from lanes l+1, l+2, ..)
        double ext[EPL + NL - 1];
#pragma unroll
        for (int e = 0; e < EPL; e++) ext[e] = (double)(T)(cur[e] * wreg[e]);
#pragma unroll
        for (int e = EPL; e < EPL + NL - 1; e++) ext[e] = from_next_lane(ext[e - EPL]);
        double part[NL];
#pragma unroll
        for (int lag = 0; lag < NL; lag++) {
            double s = 0.0;
#pragma unroll
            for (int e = 0; e < EPL; e++) s = fma(ext[e], ext[e + lag], s);
            part[lag] = s;
        }
        // Q1: r[lag] = x[0] + sum_{i>=1} x[i]x[i+lag]  ==  S[lag] - x0*x[lag] + x0 ; lane 0 owns the i = 0 term
        const double x0 = readlane_f64(ext[0], 0);
        if (x0 != 0.0) {
            if (lane == 0) {
#pragma unroll
                for (int lag = 0; lag < NL; lag++) part[lag] = (part[lag] - x0 * ext[lag]) + x0;
            }
        }
#pragma unroll
        for (int lag = 0; lag < NL; lag++) TR[lane * TS + lag] = part[lag];
        wave_sync();
#pragma unroll
        for (int lbase = 0; lbase < NL; lbase += 16) {   // 16 lags per sweep (4 lanes per lag)
            const int rl = lbase + red_lag;
            double tot = 0.0;
            if (rl < NL) {
#pragma unroll
                for (int t = 0; t < 16; t++) tot += TR[(red_part * 16 + t) * TS + rl];
            }
            tot += dpp_f64<DPP_QUAD_XOR1>(tot);
            tot += dpp_f64<0x4E>(tot);               // quad_perm [2,3,0,1]
            if (red_part == 0 && rl < NL) R[g * TS + rl] = tot;
        }
        wave_sync();
#pragma unroll
        for (int e = 0; e < EPL; e++) { cur[e] = nxt[e]; nxt[e] = nx2[e]; }
    }

    // lane f <-> frame f0 + f.  With the conditioning probe (lpc_list != NULL; f64 only): lanes FPW .. 4 FPW - 1 run the SAME
    // recursion for frame lane % FPW on lag sums moved by +-16 eps of r[0] (three sign patterns per frame: the lanes are idle
    // otherwise), and a frame whose row such a perturbation moves by more than LPC_PROBE_TOL in the parity metric goes to
    // lpc_exact_list_kernel (k_lpc_exact.hip; vbx_spectral.hpp levinson_probe has the reasoning).
    const bool probe = 4 * FPW <= 64 && sizeof(T) == 8 && lpc_list != nullptr && out_lpc != nullptr;      // uniform
    const int fi = probe ? (lane & (FPW - 1)) : lane, pat = probe ? lane / FPW : 0;
    double r[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) r[k] = (fi < nf) ? R[fi * TS + k] : 1.0;
    if (normalize) {   // Normalize::normalize over the n_lags coefficients (max |.| over all, Q2)
        double m = fabs(r[0]);
#pragma unroll
        for (int k = 1; k < NL; k++) if (k < n_lags) { const double a = fabs(r[k]); m = (a > m) ? a : m; }
        const double scale = 1.0 / m;
#pragma unroll
        for (int k = 0; k < NL; k++) r[k] = r[k] * scale;
        wave_sync();
        if (lane < FPW) {
#pragma unroll
            for (int k = 0; k < NL; k++) R[lane * TS + k] = r[k];
        }
        wave_sync();
    }
    if (out_r != nullptr) {          // [nf, n_lags] block, contiguous in the output
        T *o = out_r + f0 * (long)n_lags;
        for (int idx = lane; idx < nf * n_lags; idx += 64) {
            const int fr = idx / n_lags, k = idx - fr * n_lags;
            o[idx] = (T)R[fr * TS + k];
        }
    }
    if (out_lpc != nullptr) {        // only instantiated/called with n_lags == NL
        double ac[NL];
        if (probe && pat > 0) {
            const double d = LPC_PROBE_EPS * fabs(r[0]);
            const unsigned h = lpc_probe_hash(pat);
#pragma unroll
            for (int k = 0; k < NL; k++) r[k] = lpc_probe_lag(r[k], d, h, k);
        }
        levinson_uniform<NL - 1>(r, ac);
        wave_sync();
        if (lane < FPW) {
#pragma unroll
            for (int k = 0; k < NL; k++) R[lane * TS + k] = ac[k];
        }
        wave_sync();
        if (probe) {
            double amax = 1.0;
#pragma unroll
            for (int k = 1; k < NL; k++) { const double m = fabs(R[fi * TS + k]); amax = (m > amax) ? m : amax; }
            const double flo = 1e-6 * amax;
            bool bad = false;
#pragma unroll
            for (int k = 1; k < NL; k++) {
                const double a0 = R[fi * TS + k], m = fabs(a0);
                bad = bad || (fabs(ac[k] - a0) > LPC_PROBE_TOL * ((m > flo) ? m : flo));
            }
            const unsigned long long mask = __ballot(bad && fi < nf);
            const unsigned long long mine = (1ull << fi) | (1ull << (fi + FPW)) | (1ull << (fi + 2 * FPW)) | (1ull << (fi + 3 * FPW));
            if (lane < nf && (mask & mine) != 0) lpc_list[atomicAdd(lpc_count, 1)] = (int32_t)(f0 + lane);
        }
        T *o = out_lpc + f0 * lpc_ld;            // rows lpc_ld elements apart (NL when dense)
        for (int idx = lane; idx < nf * NL; idx += 64) {
            const int fr = idx / NL, k = idx - fr * NL;
            o[fr * lpc_ld + k] = (T)R[fr * TS + k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Many-lag path (n_lags up to frame_len): frame staged in LDS as the padded image of vbx_autocorr.hpp, lags on the
// FP64 matrix cores, 256 per tile; only the tiles that hold requested lags are computed.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void autocorr_tiles_kernel(
    const T *__restrict__ x, long n_frames, int n, long stride, const T *__restrict__ window,
    int n_lags, T *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long f = xcd_item(blockIdx.x, n_frames);
    if (f >= n_frames) return;
    const int lane = lane_id();
    const T *xf = x + f * stride;
    double *zs = smem;
    const int total = ac_mf_lds_doubles(n);
    for (int p = lane; p < total; p += 64) zs[p] = 0.0;
    wave_sync();
    for (int i = lane; i < n; i += 64) {
        double v = (double)xf[i];
        if (window != nullptr) v = (double)(T)(v * (double)window[i]);
        zs[ac_mf_phys(i)] = v;
    }
    wave_sync();
    const double x0 = zs[ac_mf_phys(0)];
    T *o = out + f * (long)n_lags;
    autocorr_mfma(zs, n, n_lags, [&](int, int lag, double s) {
        if (lag < n_lags) o[lag] = (T)((s - x0 * zs[ac_mf_phys(lag < n ? lag : n)]) + x0);   // lags >= n: empty sum, r = x0
    });
}

// Normalize::normalize on rows (src/waves.rs:68-75): one wavefront per row.
template <typename T>
__global__ __launch_bounds__(64) void normalize_rows_kernel(T *__restrict__ data, long n_rows, int n) {
    const long row = blockIdx.x;
    if (row >= n_rows) return;
    const int lane = lane_id();
    T *d = data + row * (long)n;
    // max_amplitude: fold keeps acc unless amp is strictly greater (NaN never wins)
    double m = -1.0;
    for (int i = lane; i < n; i += 64) { double a = fabs((double)d[i]); m = (a > m) ? a : m; }
    m = wave_max(m);
    const T scale = (T)1 / (T)m;                          // identity / max in the Sample's own float type (:69-71)
    for (int i = lane; i < n; i += 64) d[i] = d[i] * scale;
}

// LPC::lpc on autocorrelation rows: one thread per row (src/spectrum.rs:63-92).
// out_kc (optional, [rows, p]): the reflection coefficients lpc_mut leaves in its `kc` argument (:74).
template <typename T>
__global__ void levinson_rows_kernel(const T *__restrict__ r, long n_rows, long r_stride, int p,
                                     T *__restrict__ out, long out_ld, T *__restrict__ out_kc) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const T *rr = r + row * r_stride;
    T *aco = out + row * out_ld;
    double ac[VBX_MAX_LPC_ORDER_K + 1], tmp[VBX_MAX_LPC_ORDER_K + 1];
    double err = (double)rr[0];
    ac[0] = 1.0;
    for (int i = 1; i <= p; i++) ac[i] = 0.0;
    for (int i = 1; i <= p; i++) {
        double acc = (double)rr[i];
        for (int j = 1; j < i; j++) acc = acc + ac[j] * (double)rr[i - j];
        const double k = -acc / err;
        ac[i] = k;
        if (out_kc != nullptr) out_kc[row * (long)p + (i - 1)] = (T)k;
        for (int j = 0; j < p; j++) tmp[j] = ac[j];
        for (int j = 1; j < i; j++) ac[j] = ac[j] + k * tmp[i - j];
        err = err * (1.0 - k * k);
    }
    for (int i = 0; i <= p; i++) aco[i] = (T)ac[i];
}

// LPC::lpc on rows of lag sums that were computed from FRAMES, with the conditioning probe (round 6).
// The recursion amplifies the rounding of its lag sums by the conditioning of the frame's Toeplitz system; on oversampled speech
// (order 12-13 at 44.1 kHz) a few eps of r[0] move a coefficient that is small against the row's largest by 1e-6 .. 1e-4 of the
// parity metric (|a - b| / max(|b|, 1e-6 max |b|)) -- in the reference's own f64 result (its sequential fold carries ~sqrt(n) eps)
// more than in this one: on 6,000 frames of the 44.1 kHz fixture the ORACLE's rows are up to 3e-4 from the same recursion in long
// double on long-double lag sums, the GPU's up to 1e-5 (profiles/r06_lpc_exact_check.txt).  No f64 evaluation can agree with another
// to 1e-6 on such a row; what CAN be done is to return the exact answer.  So the recursion is repeated on lag sums moved by
// +-LPC_PROBE_EPS of r[0] (NPAT sign patterns; a lane per row: the repeats cost a few instructions per frame), and a row that such
// a perturbation moves by more than LPC_PROBE_TOL in the parity metric is listed; lpc_exact_list_kernel (k_lpc_exact.hip) redoes
// the listed frames from their samples in double-double -- the exact row rounded once.  A row that stays is within
// LPC_PROBE_TOL / 2 of the exact answer (the lag sums' own error is < 8 eps of r[0] by two transforms, ~1 eps by direct sums).
// IN PLACE is allowed (r == out, r_stride == out_ld): the fused kernels leave r[0..12] in the frame's LPC row.

// mfcc_rows != NULL: the same lane also finishes the row's MFCC (the deferred tail of MFCC::mfcc, k_mfcc.hip mfcc_row_tail: log10 + DCT of
// the num_coeffs mel filter sums the fused kernel left there) -- in the fused call's record the two rows are neighbours, one pass
// over the record's cache lines instead of two.
template <int P, bool PROBE>
__global__ __launch_bounds__(64) void levinson_rows_kernel_t(const double *__restrict__ r, long n_rows, long r_stride, double *__restrict__ out,
                                                             long out_ld, int32_t *__restrict__ lpc_list, int32_t *__restrict__ lpc_count,
                                                             double *__restrict__ mfcc_rows, long mfcc_ld, int num_coeffs,
                                                             const double *__restrict__ dct_table) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    double rr[P + 1], a0[P + 1];
#pragma unroll
    for (int k = 0; k <= P; k++) rr[k] = r[row * r_stride + k];
    if (mfcc_rows != nullptr) mfcc_row_tail(mfcc_rows + row * mfcc_ld, num_coeffs, dct_table);
    levinson_uniform<P>(rr, a0);
    double *aco = out + row * out_ld;
#pragma unroll
    for (int k = 0; k <= P; k++) aco[k] = a0[k];
    if constexpr (PROBE) {
        double amax = 1.0;
#pragma unroll
        for (int k = 1; k <= P; k++) { const double m = fabs(a0[k]); amax = (m > amax) ? m : amax; }
        const double flo = 1e-6 * amax, d = LPC_PROBE_EPS * fabs(rr[0]);
        bool bad = false;
        for (int pat = 1; pat <= LPC_PROBE_NPAT; pat++) {
            const unsigned h = lpc_probe_hash(pat);
            double rp[P + 1], ap[P + 1];
#pragma unroll
            for (int k = 0; k <= P; k++) rp[k] = lpc_probe_lag(rr[k], d, h, k);
            levinson_uniform<P>(rp, ap);
#pragma unroll
            for (int k = 1; k <= P; k++) {
                const double m = fabs(a0[k]);
                bad = bad || (fabs(ap[k] - a0[k]) > LPC_PROBE_TOL * ((m > flo) ? m : flo));      // a NaN row (the reference's NaN row) is never listed
            }
        }
        if (bad) lpc_list[atomicAdd(lpc_count, 1)] = (int32_t)row;
    }
}

// any order (runtime loops, rows in scratch): autocorrelate -> levinson of vbx_autocorr_lpc_f64 at the orders without a fused kernel
__global__ void levinson_rows_probe_kernel(const double *__restrict__ r, long n_rows, long r_stride, int p,
                                           double *__restrict__ out, long out_ld, int32_t *__restrict__ lpc_list,
                                           int32_t *__restrict__ lpc_count) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const double *rr = r + row * r_stride;
    double *aco = out + row * out_ld;
    double ac[VBX_MAX_LPC_ORDER_K + 1], tmp[VBX_MAX_LPC_ORDER_K + 1], r0[VBX_MAX_LPC_ORDER_K + 1];
    for (int i = 0; i <= p; i++) r0[i] = rr[i];             // (in place: the row is read before it is written)
    bool bad = false;
    double amax = 1.0;
    for (int pat = 0; pat <= LPC_PROBE_NPAT; pat++) {
        const double d = pat ? LPC_PROBE_EPS * fabs(r0[0]) : 0.0;
        const unsigned h = lpc_probe_hash(pat);
        auto lag = [&](int k) { return pat ? lpc_probe_lag(r0[k], d, h, k) : r0[k]; };
        double err = lag(0);
        ac[0] = 1.0;
        for (int i = 1; i <= p; i++) ac[i] = 0.0;
        for (int i = 1; i <= p; i++) {
            double acc = lag(i);
            for (int j = 1; j < i; j++) acc = acc + ac[j] * lag(i - j);
            const double k = -acc / err;
            ac[i] = k;
            for (int j = 0; j < p; j++) tmp[j] = ac[j];
            for (int j = 1; j < i; j++) ac[j] = ac[j] + k * tmp[i - j];
            err = err * (1.0 - k * k);
        }
        if (pat == 0) {
            for (int i = 0; i <= p; i++) { aco[i] = ac[i]; const double m = fabs(ac[i]); amax = (m > amax) ? m : amax; }
        } else {
            const double flo = 1e-6 * amax;
            for (int i = 1; i <= p; i++) {
                const double a0 = aco[i], m = fabs(a0);
                bad = bad || (fabs(ac[i] - a0) > LPC_PROBE_TOL * ((m > flo) ? m : flo));
            }
        }
    }
    if (bad) lpc_list[atomicAdd(lpc_count, 1)] = (int32_t)row;
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------

template <int EPL, int NL, typename T>
static void launch_fewlags(hipStream_t s, const T *x, long F, int n, long stride, const T *window,
                           int n_lags, int normalize, T *out_r, T *out_lpc, long lpc_ld, int32_t *lpc_list, int32_t *lpc_count) {
    hipLaunchKernelGGL((autocorr_fewlags_kernel<EPL, NL, T>), dim3((unsigned)((F + AC_FPW - 1) / AC_FPW)), dim3(64), 0, s,
                       x, F, n, stride, window, n_lags, normalize, out_r, out_lpc, lpc_ld, lpc_list, lpc_count);
}

template <int NL, typename T>
static bool dispatch_epl(hipStream_t s, const T *x, long F, int n, long stride, const T *window,
                         int n_lags, int normalize, T *out_r, T *out_lpc, long lpc_ld, int32_t *lpc_list = nullptr, int32_t *lpc_count = nullptr) {
    if (n <= 64 * 8) launch_fewlags<8, NL, T>(s, x, F, n, stride, window, n_lags, normalize, out_r, out_lpc, lpc_ld, lpc_list, lpc_count);
    else if (n <= 64 * 16) launch_fewlags<16, NL, T>(s, x, F, n, stride, window, n_lags, normalize, out_r, out_lpc, lpc_ld, lpc_list, lpc_count);
    else if (n <= 64 * 20) launch_fewlags<20, NL, T>(s, x, F, n, stride, window, n_lags, normalize, out_r, out_lpc, lpc_ld, lpc_list, lpc_count);
    else if (n <= 64 * 32) launch_fewlags<32, NL, T>(s, x, F, n, stride, window, n_lags, normalize, out_r, out_lpc, lpc_ld, lpc_list, lpc_count);
    else return false;
    return true;
}

// true if a register-resident few-lag kernel exists for this shape
bool fewlags_supported(int n, int n_lags, bool want_lpc) {
    if (n > 64 * 32 || n < 1) return false;
    if (want_lpc) return n_lags == 9 || n_lags == 11 || n_lags == 13 || n_lags == 17;
    return n_lags <= 17;
}

template <typename T>
static void launch_autocorr_fewlags_t(hipStream_t s, const T *x, long F, int n, long stride, const T *window,
                                      int n_lags, int normalize, T *out_r, T *out_lpc, long lpc_ld, int32_t *lpc_list = nullptr,
                                      int32_t *lpc_count = nullptr) {
    if (lpc_ld <= 0) lpc_ld = n_lags;
    if (out_lpc != nullptr) {
        switch (n_lags) {
            case 9:  dispatch_epl<9, T>(s, x, F, n, stride, window, n_lags, normalize, out_r, out_lpc, lpc_ld, lpc_list, lpc_count); break;
            case 11: dispatch_epl<11, T>(s, x, F, n, stride, window, n_lags, normalize, out_r, out_lpc, lpc_ld, lpc_list, lpc_count); break;
            case 13: dispatch_epl<13, T>(s, x, F, n, stride, window, n_lags, normalize, out_r, out_lpc, lpc_ld, lpc_list, lpc_count); break;
            case 17: dispatch_epl<17, T>(s, x, F, n, stride, window, n_lags, normalize, out_r, out_lpc, lpc_ld, lpc_list, lpc_count); break;
        }
        return;
    }
    if (n_lags <= 9) dispatch_epl<9, T>(s, x, F, n, stride, window, n_lags, normalize, out_r, (T *)nullptr, lpc_ld);
    else if (n_lags <= 13) dispatch_epl<13, T>(s, x, F, n, stride, window, n_lags, normalize, out_r, (T *)nullptr, lpc_ld);
    else dispatch_epl<17, T>(s, x, F, n, stride, window, n_lags, normalize, out_r, (T *)nullptr, lpc_ld);
}

void launch_autocorr_fewlags(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                             int n_lags, int normalize, double *out_r, double *out_lpc, long lpc_ld, int32_t *lpc_list, int32_t *lpc_count) {
    launch_autocorr_fewlags_t<double>(s, x, F, n, stride, window, n_lags, normalize, out_r, out_lpc, lpc_ld, lpc_list, lpc_count);
}
void launch_autocorr_fewlags_f32(hipStream_t s, const float *x, long F, int n, long stride, const float *window,
                                 int n_lags, int normalize, float *out_r, float *out_lpc, long lpc_ld) {
    launch_autocorr_fewlags_t<float>(s, x, F, n, stride, window, n_lags, normalize, out_r, out_lpc, lpc_ld);
}

void launch_autocorr_tiles(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                           int n_lags, double *out) {
    const size_t lds = (size_t)ac_mf_lds_doubles(n) * sizeof(double);
    hipLaunchKernelGGL(autocorr_tiles_kernel<double>, dim3((unsigned)F), dim3(64), lds, s, x, F, n, stride, window, n_lags, out);
}
void launch_autocorr_tiles_f32(hipStream_t s, const float *x, long F, int n, long stride, const float *window,
                               int n_lags, float *out) {
    const size_t lds = (size_t)ac_mf_lds_doubles(n) * sizeof(double);
    hipLaunchKernelGGL(autocorr_tiles_kernel<float>, dim3((unsigned)F), dim3(64), lds, s, x, F, n, stride, window, n_lags, out);
}

void launch_normalize_rows(hipStream_t s, double *data, long rows, int n) {
    hipLaunchKernelGGL(normalize_rows_kernel<double>, dim3((unsigned)rows), dim3(64), 0, s, data, rows, n);
}
void launch_normalize_rows_f32(hipStream_t s, float *data, long rows, int n) {
    hipLaunchKernelGGL(normalize_rows_kernel<float>, dim3((unsigned)rows), dim3(64), 0, s, data, rows, n);
}

void launch_levinson_rows(hipStream_t s, const double *r, long rows, long r_stride, int p, double *out, long out_ld,
                          double *out_kc) {
    const int bs = 64;
    hipLaunchKernelGGL(levinson_rows_kernel<double>, dim3((unsigned)((rows + bs - 1) / bs)), dim3(bs), 0, s, r, rows, r_stride, p, out, out_ld, out_kc);
}
void launch_levinson_rows_probe(hipStream_t s, const double *r, long rows, long r_stride, int p, double *out, long out_ld,
                                int32_t *lpc_list, int32_t *lpc_count, double *mfcc_rows, long mfcc_ld, int num_coeffs, const double *dct) {
    const int bs = 64;
    const dim3 grid((unsigned)((rows + bs - 1) / bs));
    if (p == 12) {                                           // the fused kernels' order: rows in registers
        if (lpc_list != nullptr) hipLaunchKernelGGL((levinson_rows_kernel_t<12, true>), grid, dim3(bs), 0, s, r, rows, r_stride, out, out_ld, lpc_list, lpc_count,
                                                    mfcc_rows, mfcc_ld, num_coeffs, dct);
        else hipLaunchKernelGGL((levinson_rows_kernel_t<12, false>), grid, dim3(bs), 0, s, r, rows, r_stride, out, out_ld, lpc_list, lpc_count,
                                mfcc_rows, mfcc_ld, num_coeffs, dct);
        return;
    }
    if (lpc_list == nullptr) { launch_levinson_rows(s, r, rows, r_stride, p, out, out_ld, nullptr); return; }
    hipLaunchKernelGGL(levinson_rows_probe_kernel, dim3((unsigned)((rows + bs - 1) / bs)), dim3(bs), 0, s, r, rows, r_stride, p, out, out_ld,
                       lpc_list, lpc_count);
}
void launch_levinson_rows_f32(hipStream_t s, const float *r, long rows, long r_stride, int p, float *out, long out_ld,
                              float *out_kc) {
    const int bs = 64;
    hipLaunchKernelGGL(levinson_rows_kernel<float>, dim3((unsigned)((rows + bs - 1) / bs)), dim3(bs), 0, s, r, rows, r_stride, p, out, out_ld, out_kc);
}

}  // namespace vbx
