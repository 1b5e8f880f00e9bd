// k_pitch.hip -- Pitched::pitch (Boersma-style autocorrelation pitch candidates).
//
// Reference: src/periodic.rs:29-87 (interpolate_sinc), :103-188 (brent_maximize),
//            :192-229 (improve_extremum), :362-375 (local_maxima), :396-455 (pitch),
//            src/waves.rs:44-75 (max_amplitude / normalize).  Quirks Q1-Q10 reproduced.
//
// One kernel, one wavefront per frame: (1) the windowed frame is staged in LDS and the all-lag autocorrelation
// runs on the FP64 matrix cores (vbx_autocorr.hpp), giving the normalised / lag-window-divided curve y, which
// replaces the frame in LDS; (2) the candidate peaks are compacted in index order, bounded from above (one lane
// per candidate) and refined best-bound-first, one candidate at a time: every lane runs the reference's Brent
// iteration on identical values (wave sums are bit-identical in all lanes), while each sinc evaluation -- the
// dominant cost, 2*(depth+1) terms -- is spread over the 64 lanes and reduced with DPP.  Candidates that provably
// cannot reach the returned top-kmax are skipped (exact).  The path is FP64 bound (hundreds of flop per byte);
// HBM traffic is the frame's samples in and 16 bytes per kept candidate out.
#include "vbx_autocorr.hpp"
#include "vbx_kernels.hpp"
#include "vbx_pitch_refine.hpp"

namespace vbx {

// ------------------------------------------------------------------------------------------
// Pitched::pitch, one wavefront per frame, one kernel (src/periodic.rs:396-455):
//  1) the windowed frame is staged in LDS as the padded image of vbx_autocorr.hpp and
//     r = self.autocorrelate(self.len()) (:403) runs on the FP64 matrix cores;
//  2) y[i] = (r[i] / max|r|) / w_lag[i] (:404-408) replaces the image in LDS (zero padded, standing for
//     resize(2N, 0), :411);
//  a) peak scan, lane-parallel: strict local maxima of y[0..N/2) (Q4), the "parabolic" lag (Q5) and
//     the frequency filter (:439); survivors are compacted in index order into an LDS list.
//     The sinc(30) strength of :433 is dead in the reference (overwritten at :448 for every
//     candidate that passes the filter, dropped otherwise) and is not evaluated.
//  b, c) refinement (improve_extremum_sinc) and the sorted candidate list, see below.
// Phase 1 keeps the matrix pipe busy and phases a-c the vector ALU; the wavefronts resident on a SIMD are in
// different phases of different frames, so both pipes work at the same time.
// ------------------------------------------------------------------------------------------
// ALIAS: a single autocorrelation pass (n <= AC_MF_NT * 256): the lag values wait in registers while y takes the
// image's place in LDS.  Otherwise y has its own region.
template <bool ALIAS>
__device__ __forceinline__ void pitch_frame_mfma(
    double *smem, const long f, const double *__restrict__ frames, int n, long stride, const double *__restrict__ window,
    const double *__restrict__ lag_window, double sample_rate, double threshold, double fmin, double fmax,
    int kmax, int full_off, double *__restrict__ out_cand, long cand_ld, int32_t *__restrict__ out_count,
    int32_t *__restrict__ status, unsigned long long *__restrict__ work, const bool pcm = false) {
    const int lane = lane_id();
    double *zs = smem;                              // padded image of the windowed frame (vbx_autocorr.hpp)
    // refinement state (vbx_pitch_refine.hpp): y[n + Y_PAD] | p16 | keys | candidate list
    double *ys = ALIAS ? smem : smem + ((ac_mf_lds_doubles(n) + 1) & ~1);
    {
        // all of the frame's loads are issued before the first LDS store (nothing else hides their latency here);
        // only the zero margins of the image are cleared, the pad double inside each 16 samples is never read
        const double *xf = frames + f * stride;
        const int16_t *x16 = reinterpret_cast<const int16_t *>(frames) + f * stride;        // the frame when pcm
        constexpr int NB = 8;
        for (int p = lane; p < ac_mf_phys(0); p += 64) zs[p] = 0.0;
        for (int p = ac_mf_phys(n) + lane; p < ac_mf_lds_doubles(n); p += 64) zs[p] = 0.0;
        for (int i0 = 0; i0 < n; i0 += 64 * NB) {
            double xv[NB], wv[NB];
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int i = i0 + 64 * j + lane;
                xv[j] = (i < n) ? (pcm ? pcm16_value(x16[i]) : xf[i]) : 0.0;
                wv[j] = (window != nullptr && i < n) ? window[i] : 1.0;
            }
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int i = i0 + 64 * j + lane;
                if (i < n) zs[ac_mf_phys(i)] = (window != nullptr) ? xv[j] * wv[j] : xv[j];
            }
        }
        wave_sync();
    }
    const double x0 = zs[ac_mf_phys(0)];
    double amax = -1.0;                             // max_amplitude over ALL lags (Q2; NaN never wins)
    if (ALIAS) {
        double rv[AC_MF_NT * 4];
        autocorr_mfma(zs, n, n, [&](int slot, int lag, double s) {          // self.autocorrelate(self.len()), :403
            const double r = (lag < n) ? (s - x0 * zs[ac_mf_phys(lag)]) + x0 : 0.0;
            rv[slot] = r;
            const double a = fabs(r);
            amax = (lag < n && a > amax) ? a : amax;
        });
        wave_sync();                                // every lane is done with the image
        amax = wave_max(amax);
        const double scale = 1.0 / amax;            // normalize (:404), then / lag window (:406-408)
#pragma unroll
        for (int slot = 0; slot < AC_MF_NT * 4; slot++) {
            const int lag = (slot >> 2) * AC_MF_TILE + 64 * (slot & 3) + lane;
            if (lag < n) ys[lag] = (rv[slot] * scale) / lag_window[lag];
        }
        if (lane < Y_PAD) ys[n + lane] = 0.0;
    } else {
        autocorr_mfma(zs, n, n, [&](int, int lag, double s) {
            if (lag < n) {
                const double r = (s - x0 * zs[ac_mf_phys(lag)]) + x0;
                ys[lag] = r;
                const double a = fabs(r);
                amax = (a > amax) ? a : amax;
            }
        });
        wave_sync();
        amax = wave_max(amax);
        const double scale = 1.0 / amax;
        for (int i = lane; i < n; i += 64) ys[i] = (ys[i] * scale) / lag_window[i];
        if (lane < Y_PAD) ys[n + lane] = 0.0;
    }
    wave_sync();

    pitch_params_t pp;
    pp.sample_rate = sample_rate; pp.threshold = threshold; pp.fmin = fmin; pp.fmax = fmax; pp.kmax = kmax; pp.full_off = full_off; pp.f32 = 0;
    double2 *full = full_off ? reinterpret_cast<double2 *>(reinterpret_cast<char *>(smem) + full_off) : nullptr;
    pitch_refine_store(ys, n, pp, f, out_cand, cand_ld, out_count, status, work, 0.0, full);
}

template <bool ALIAS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void pitch_kernel(
    const double *__restrict__ frames, long n_frames, int n, long stride, const double *__restrict__ window,
    const double *__restrict__ lag_window, double sample_rate, double threshold, double fmin, double fmax,
    int kmax, int full_off, double *__restrict__ out_cand, long cand_ld, int32_t *__restrict__ out_count,
    int32_t *__restrict__ status, unsigned long long *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long f = xcd_item(blockIdx.x, n_frames);          // neighbouring frames on the same XCD: their overlap hits its L2
    if (f >= n_frames) return;
    pitch_frame_mfma<ALIAS>(smem, f, frames, n, stride, window, lag_window, sample_rate, threshold, fmin, fmax, kmax,
                            full_off, out_cand, cand_ld, out_count, status, work);
}

// The same per-frame routine over a list of frame indices: the frames the FFT-based kernel (k_spectral.hip) could not
// decide within its rounding error.  A fixed grid walks the list; *list_count is read on the device.
template <bool ALIAS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void pitch_list_kernel(
    const int32_t *__restrict__ frame_list, const int32_t *__restrict__ list_count,
    const double *__restrict__ frames, int n, long stride, const double *__restrict__ window,
    const double *__restrict__ lag_window, double sample_rate, double threshold, double fmin, double fmax,
    int kmax, int full_off, double *__restrict__ out_cand, long cand_ld, int32_t *__restrict__ out_count,
    int32_t *__restrict__ status, unsigned long long *__restrict__ work, int pcm) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int count = *list_count;
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        pitch_frame_mfma<ALIAS>(smem, (long)frame_list[i], frames, n, stride, window, lag_window, sample_rate, threshold,
                                fmin, fmax, kmax, full_off, out_cand, cand_ld, out_count, status, work, pcm != 0);
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------
// interpolate_sinc / improve_extremum at M query points of one curve (PG lanes per point)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sinc_points_kernel(const double *__restrict__ y, int ylen, int offset, int nx,
                                                         const double *__restrict__ xs, long m, int depth,
                                                         double *__restrict__ out, int32_t *__restrict__ status) {
    const long q = (long)blockIdx.x * PNG + lane_id() / PG;
    const bool have = q < m;
    int st = 0;
    const double v = sinc_interp<PG>(y, ylen, ylen, offset, nx, have ? xs[q] : 0.0, depth, have, st);
    if (have && (lane_id() & (PG - 1)) == 0) {
        out[q] = (st & 4) ? 0.0 : v;
        if (status != nullptr) status[q] = (st & 4) ? 4 : 0;
    }
}

// interp: 0 Interpolation::None, 1 Parabolic, 2 Sinc(depth) (src/periodic.rs:192-229); is_max == 0 negates the interpolant
__global__ __launch_bounds__(64) void extremum_points_kernel(const double *__restrict__ y, int ylen, int offset, int nx,
                                                             const double *__restrict__ ix, long m, int depth,
                                                             double *__restrict__ out_xy, int32_t *__restrict__ status,
                                                             int interp, int is_max) {
    const long q = (long)blockIdx.x * PNG + lane_id() / PG;
    const bool have = q < m;
    int st = 0;
    double xmid = 0., ymid = 0.;
    if (interp == 2) {
        improve_extremum_sinc<PG>(y, ylen, ylen, offset, nx, have ? ix[q] : 1.0, depth, have, xmid, ymid, st, nullptr, nullptr,
                                  -__builtin_inf(), nullptr, is_max == 0);
    } else if (have) {
        const double ixmid = ix[q];
        if (ylen < 1) st |= 4;
        else if (ixmid == 0.) { xmid = 0.; ymid = y[0]; }                                              // :193
        else if (ixmid >= (double)nx) {                                                                // :194
            if (nx < 1 || nx - 1 >= ylen) st |= 4; else { xmid = (double)nx; ymid = y[nx - 1]; }
        } else if (interp == 0) { xmid = 0.; ymid = y[0]; }                                            // :197-199
        else {                                                                                         // :200-207
#pragma clang fp contract(off)
            const double fl = floor(ixmid);
            if (!(fl >= 1.) || !(fl + 1. < (double)ylen)) st |= 4;       // usize underflow / index out of bounds (NaN too)
            else {
                const int i = (int)fl;
                const double diff = y[i + 1] - y[i - 1];
                const double mid = y[i];
                const double dy = 0.5 * diff;
                const double d2y = 2.0 * mid - diff;
                xmid = ixmid + dy / d2y;
                ymid = mid + 0.5 * dy * dy / d2y;
            }
        }
    }
    if (have && (lane_id() & (PG - 1)) == 0) {
        out_xy[2 * q] = (st & 4) ? 0.0 : xmid;
        out_xy[2 * q + 1] = (st & 4) ? 0.0 : ymid;
        if (status != nullptr) status[q] = (st & 4) ? 4 : 0;
    }
}

size_t pitch_lds_bytes(int n) {
    const int nblk = (n + PB - 1) / PB;
    const size_t refine = (size_t)(n + Y_PAD + ((nblk + 2) & ~1)) * sizeof(double) + (size_t)(n / 4 + 8) * (sizeof(float) + sizeof(cand_t));
    const size_t image = (size_t)((ac_mf_lds_doubles(n) + 1) & ~1) * sizeof(double);
    if (n <= AC_MF_NT * AC_MF_TILE) return image > refine ? image : refine;
    return image + refine;
}

// kmax > PITCH_LIST_LANES: the whole candidate Vec is wanted; its entries are parked in an extra LDS region behind
// the frame state (vbx_pitch_refine.hpp, `full`) and rank-sorted at the end of the frame
size_t pitch_full_list_bytes(int n, int kmax) {
    return kmax > PITCH_LIST_LANES ? (size_t)pitch_full_list_entries(n) * sizeof(double2) : 0;
}

void launch_pitch(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                  const double *lag_window, double sample_rate, double threshold, double fmin, double fmax,
                  int kmax, pitch_t *out_cand, long cand_ld, int32_t *out_count, int32_t *status,
                  unsigned long long *work) {
    const size_t base = (pitch_lds_bytes(n) + 15) & ~(size_t)15, extra = pitch_full_list_bytes(n, kmax);
    const int full_off = extra ? (int)base : 0;
    if (n <= AC_MF_NT * AC_MF_TILE)
        hipLaunchKernelGGL((pitch_kernel<true>), dim3((unsigned)F), dim3(64), base + extra, s,
                           x, F, n, stride, window, lag_window, sample_rate, threshold, fmin, fmax, kmax, full_off,
                           reinterpret_cast<double *>(out_cand), cand_ld, out_count, status, work);
    else
        hipLaunchKernelGGL((pitch_kernel<false>), dim3((unsigned)F), dim3(64), base + extra, s,
                           x, F, n, stride, window, lag_window, sample_rate, threshold, fmin, fmax, kmax, full_off,
                           reinterpret_cast<double *>(out_cand), cand_ld, out_count, status, work);
}

void launch_pitch_list(hipStream_t s, const int32_t *frame_list, const int32_t *list_count, int grid,
                       const double *x, int n, long stride, const double *window,
                       const double *lag_window, double sample_rate, double threshold, double fmin, double fmax,
                       int kmax, pitch_t *out_cand, long cand_ld, int32_t *out_count, int32_t *status,
                       unsigned long long *work, bool pcm) {
    const size_t base = (pitch_lds_bytes(n) + 15) & ~(size_t)15, extra = pitch_full_list_bytes(n, kmax);
    const int full_off = extra ? (int)base : 0;
    if (n <= AC_MF_NT * AC_MF_TILE)
        hipLaunchKernelGGL((pitch_list_kernel<true>), dim3((unsigned)grid), dim3(64), base + extra, s,
                           frame_list, list_count, x, n, stride, window, lag_window, sample_rate, threshold, fmin, fmax, kmax,
                           full_off, reinterpret_cast<double *>(out_cand), cand_ld, out_count, status, work, pcm ? 1 : 0);
    else
        hipLaunchKernelGGL((pitch_list_kernel<false>), dim3((unsigned)grid), dim3(64), base + extra, s,
                           frame_list, list_count, x, n, stride, window, lag_window, sample_rate, threshold, fmin, fmax, kmax,
                           full_off, reinterpret_cast<double *>(out_cand), cand_ld, out_count, status, work, pcm ? 1 : 0);
}

void launch_sinc_points(hipStream_t s, const double *y, int ylen, long offset, long nx, const double *xs, long m,
                        long depth, double *out, int32_t *status) {
    hipLaunchKernelGGL(sinc_points_kernel, dim3((unsigned)((m + PNG - 1) / PNG)), dim3(64), 0, s,
                       y, ylen, (int)offset, (int)nx, xs, m, (int)depth, out, status);
}

void launch_extremum_points(hipStream_t s, const double *y, int ylen, long offset, long nx, const double *ix, long m,
                            long depth, double *out_xy, int32_t *status, int interp, int is_max) {
    hipLaunchKernelGGL(extremum_points_kernel, dim3((unsigned)((m + PNG - 1) / PNG)), dim3(64), 0, s,
                       y, ylen, (int)offset, (int)nx, ix, m, (int)depth, out_xy, status, interp, is_max);
}

}  // namespace vbx
