// k_pitch.hip -- Pitched::pitch (Boersma-style autocorrelation pitch candidates).
//
// Reference: src/periodic.rs:29-87 (interpolate_sinc), :103-188 (brent_maximize),
//            :192-229 (improve_extremum), :362-375 (local_maxima), :396-455 (pitch),
//            src/waves.rs:44-75 (max_amplitude / normalize).  Quirks Q1-Q10 reproduced.
//
// One wavefront per frame.  The windowed frame is staged in LDS, the all-lag autocorrelation
// runs as lag tiles (vbx_autocorr.hpp), the normalised / lag-window-divided curve y stays in
// LDS, and every candidate peak is refined by the reference's Brent iteration whose scalar
// control flow is executed identically by all 64 lanes while each sinc evaluation -- the
// dominant cost, 2*(depth+1) terms -- is spread over the lanes and reduced with DPP.
//
// Sinc term algebra (exact identities, no change of the reference's formula):
//   sin(pi*(phi+n)) = (-1)^n * sin(pi*phi)                       -> one sinpi per evaluation
//   cos(a/(phi+D)) with a/(phi+D) in [0, pi]                      -> odd polynomial in (theta - pi/2)
//   1/a                                                            -> v_rcp_f64 + 2 Newton steps
// This kernel is FP64-VALU bound (hundreds of flop per byte); HBM traffic is the 3.8 KB of
// new samples per frame.
#include "vbx_autocorr.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

// reciprocal of a normal, well-scaled double
__device__ __forceinline__ double fast_rcp(double a) {
    double r = __builtin_amdgcn_rcp(a);
    r = fma(fma(-a, r, 1.0), r, r);
    r = fma(fma(-a, r, 1.0), r, r);
    return r;
}

// cos(theta) for theta in [0, pi] (slightly outside is fine): -sin(theta - pi/2), Taylor to u^21
__device__ __forceinline__ double cos_0_pi(double theta) {
    const double u = theta - 1.57079632679489661923;
    const double u2 = u * u;
    double p = -1.9572941063391261231e-20;           // -1/21!
    p = fma(p, u2, 8.2206352466243297170e-18);       //  1/19!
    p = fma(p, u2, -2.8114572543455207632e-15);      // -1/17!
    p = fma(p, u2, 7.6471637318198164759e-13);       //  1/15!
    p = fma(p, u2, -1.6059043836821614599e-10);      // -1/13!
    p = fma(p, u2, 2.5052108385441718775e-08);       //  1/11!
    p = fma(p, u2, -2.7557319223985890653e-06);      // -1/9!
    p = fma(p, u2, 1.9841269841269841270e-04);       //  1/7!
    p = fma(p, u2, -8.3333333333333333333e-03);      // -1/5!
    p = fma(p, u2, 1.6666666666666666667e-01);       //  1/3!   (sign folded below)
    // sin(u) = u - u^3/6 + ... = u * (1 - u2*(1/6 - u2*(1/120 - ...)))
    const double s = u * fma(-u2, p, 1.0);
    return -s;
}

// y lookup: entries [nstore, ylen) are the zeros of self_lag.resize(2N, 0) (src/periodic.rs:411)
__device__ __forceinline__ double y_at(const double *y, int nstore, long idx) {
    return (idx < nstore) ? y[idx] : 0.0;
}

// interpolate_sinc, wave-cooperative; result identical in all lanes.  st |= 4 where the
// reference would index out of bounds.
__device__ __forceinline__ double sinc_interp(const double *y, int nstore, long ylen, long offset, long nx,
                                              double x, long max_depth, int &st) {
    if (nx < 1) return __builtin_nan("");                                     // :38
    if (x > (double)nx) {                                                      // :39
        const long idx = offset + nx - 1;
        if (idx < 0 || idx >= ylen) { st |= 4; return 0.0; }
        return y_at(y, nstore, idx);
    }
    if (x < 0.0) return y_at(y, nstore, 0);                                   // :40
    const double fl = floor(x);
    const long nl = (fl > 0.0) ? (long)fl : 0;                                // NaN -> 0
    const long nr = nl + 1;
    const double phil = x - (double)nl;
    const double phir = 1.0 - phil;
    if (fabs(x - (double)nl) < 1.0e-10) {                                      // :41
        const long idx = offset + nl;
        if (idx < 0 || idx >= ylen) { st |= 4; return 0.0; }
        return y_at(y, nstore, idx);
    }
    if (fabs(x - (double)nr) < 1.0e-10) {                                      // :42
        const long idx = offset + nr;
        if (idx < 0 || idx >= ylen) { st |= 4; return 0.0; }
        return y_at(y, nstore, idx);
    }
    if ((offset + nr) < max_depth) max_depth = ((offset + nr) < 0) ? 0 : (offset + nr);    // :46-52
    if ((offset + nl + max_depth) >= nx) {                                                    // :55-57
        max_depth = nx - offset + nl - 1;
        if (max_depth < 0) { st |= 4; return 0.0; }
    }
    if (offset + nr >= ylen) { st |= 4; return 0.0; }    // left index at n = 0 (:67) out of bounds

    const int lane = lane_id();
    const int side = lane & 1;                 // even lanes: "left" terms, odd lanes: "right" terms
    const double ph = side ? phir : phil;
    const double s0 = sinpi(ph);               // sin(pi*(ph+n)) = (-1)^n * s0
    const double dd = ph + (double)max_depth;
    const double inv_dd = 1.0 / dd;
    const long ibase = side ? (offset + nl) : (offset + nr);
    const long nterms = max_depth + 1;
    double acc = 0.0;
    for (long n = (lane >> 1); n < nterms; n += 32) {
        const double a = M_PI * (ph + (double)n);
        long idx = side ? (ibase + n) : (ibase - n);
        idx = (idx < 0) ? 0 : idx;
        idx = (idx >= ylen) ? (ylen - 1) : idx;          // only reachable on the right side (:78)
        const double r_lag = y_at(y, nstore, idx);
        const double sgn_s0 = (n & 1) ? -s0 : s0;
        const double first = sgn_s0 * fast_rcp(a);
        const double second = fma(0.5, cos_0_pi(a * inv_dd), 0.5);
        acc = fma(r_lag * first, second, acc);
    }
    return wave_sum(acc);
}

// brent_maximize (src/periodic.rs:103-188) over f(x) = interpolate_sinc(.., x, depth): a MINIMISER
// of the un-negated interpolant (Q8).  All lanes run the same scalar iteration.
__device__ __forceinline__ double brent_sinc(const double *y, int nstore, long ylen, long offset, long nx, long depth,
                                             double a, double b, double tol, double &fx, int &st) {
#pragma clang fp contract(off)   // keep the scalar iteration bit-identical to the unfused CPU arithmetic
    const double golden = 1. - 0.6180339887498948482045868343656381177203091798057628621;
    const double sqrt_epsilon = 1.4901161193847656e-08;   // sqrt(f64::EPSILON)
    const double eps = 2.220446049250313e-16;
    double v = a + golden * (b - a);
    double fv = sinc_interp(y, nstore, ylen, offset, nx, v, depth, st);
    double x = v, w = v;
    fx = fv;
    double fw = fv;
    for (int it = 1; it <= 60; it++) {
        const double range = b - a;
        const double middle_range = (a + b) * 0.5;
        const double tol_act = sqrt_epsilon * fabs(x) + tol / 3.;
        if (fabs(x - middle_range) + range * 0.5 <= 2. * tol_act) return x;
        double new_step = (x < middle_range) ? golden * (b - x) : golden * (a - x);
        if (fabs(x - w) >= tol_act) {
            const double t = (x - w) * (fx - fv);
            double q = (x - v) * (fx - fw);
            double p = (x - v) * q - (x - w) * t;
            q = 2. * q - t;
            if (q > 0.) p = -p; else q = -q;
            if (fabs(p) < fabs(new_step * q) && p > q * (a - x + 2. * tol_act) && p < q * (b - x - 2. * tol_act))
                new_step = p / q;
        }
        if (fabs(new_step) < tol_act) new_step = (new_step > 0.) ? tol_act : -tol_act;
        const double t = x + new_step;
        const double ft = sinc_interp(y, nstore, ylen, offset, nx, t, depth, st);
        if (ft <= fx) {
            if (t < x) b = x; else a = x;
            v = w; w = x; x = t;
            fv = fw; fw = fx; fx = ft;
        } else {
            if (t < x) a = t; else b = t;
            if (ft <= fw || fabs(w - x) < eps) {
                v = w; w = t;
                fv = fw; fw = ft;
            } else if (ft <= fv || fabs(v - x) < eps || fabs(v - w) < eps) {
                v = t;
                fv = ft;
            }
        }
    }
    return x;
}

// improve_extremum(.., Sinc(depth), true), src/periodic.rs:192-229
__device__ __forceinline__ void improve_extremum_sinc(const double *y, int nstore, long ylen, long offset, long nx,
                                                      double ixmid, long depth, double &xmid, double &ymid, int &st) {
    if (ixmid == 0.) { xmid = 0.; ymid = y_at(y, nstore, 0); return; }                        // :193
    if (ixmid >= (double)nx) {                                                                  // :194
        if (nx < 1 || nx - 1 >= ylen) { st |= 4; xmid = 0.; ymid = 0.; return; }
        xmid = (double)nx; ymid = y_at(y, nstore, nx - 1); return;
    }
    const double a = ixmid - 1., b = ixmid + 1.;
    if (!(a < b)) { st |= 4; xmid = 0.; ymid = 0.; return; }                                   // assert!(a < b), :113
    double fx = 0.;
    xmid = brent_sinc(y, nstore, ylen, offset, nx, depth, a, b, 1e-10, fx, st);
    ymid = fx;
}

// ------------------------------------------------------------------------------------------
// pitch kernel
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pitch_kernel(
    const double *__restrict__ x, long n_frames, int n, long stride, const double *__restrict__ window,
    const double *__restrict__ lag_window, double sample_rate, double threshold, double fmin, double fmax,
    int kmax, pitch_t *__restrict__ out_cand, int32_t *__restrict__ out_count, int32_t *__restrict__ status) {
    extern __shared__ double smem[];
    const long f = blockIdx.x;
    if (f >= n_frames) return;
    const int lane = lane_id();
    double *xs = smem;                              // [n + pad] windowed samples, zero padded
    double *ys = smem + n + autocorr_pad(n);        // [n] lag curve
    const double *xf = x + f * stride;
    const int total = n + autocorr_pad(n);
    for (int i = lane; i < total; i += 64) {
        double v = 0.0;
        if (i < n) { v = xf[i]; if (window != nullptr) v *= window[i]; }
        xs[i] = v;
    }
    __syncthreads();

    // self.autocorrelate(self.len()), :403
    const double x0 = xs[0];
    double amax = -1.0;
    autocorr_tiles(xs, n, n, [&](int lag, double s) {
        const double r = (s - x0 * xs[lag]) + x0;
        ys[lag] = r;
        const double a = fabs(r);
        amax = (a > amax) ? a : amax;
    });
    amax = wave_max(amax);                          // max_amplitude over ALL lags (Q2)
    __syncthreads();
    // normalize (:404) then divide by the lag window (:406-408)
    const double scale = 1.0 / amax;
    for (int i = lane; i < n; i += 64) ys[i] = (ys[i] * scale) / lag_window[i];
    __syncthreads();

    const long b = (long)floor(0.5 * (double)n);    // brent_ixmax, :414
    const long offset = -b - 1;                     // :429
    const long nx = b - offset;                     // :430
    const long ylen = 2L * n;                       // :411

    int st = 0;
    int total_cand = 0, kept = 0;
    double lf = 0.0, ls = 0.0;                      // lane j holds sorted candidate j
    bool any_nan = false;

    for (long base = 0; base < b; base += 64) {
        const long k = base + lane;
        bool ispeak = false;
        if (k >= 1 && k + 1 < b) {                  // windows(3) over self_lag[0..b] (Q4)
            const double c = ys[k];
            ispeak = (ys[k - 1] < c) && (ys[k + 1] < c);
        }
        unsigned long long mask = __ballot(ispeak);
        while (mask) {
            const int bit = __builtin_ctzll(mask);
            mask &= mask - 1;
            const long kk = base + bit;
            const double peak = ys[kk], peak_rev = ys[kk - 1], peak_fwd = ys[kk + 1];
            const double dr = 0.5 * (peak_fwd - peak_rev);                    // :423
            const double d2r = 2. * peak - (peak_rev - peak_fwd);             // :424 (Q5)
            const double freq = sample_rate / ((double)kk + dr / d2r);        // :425
            const double nn = sample_rate / freq - (double)offset;            // :432
            double strn = sinc_interp(ys, n, ylen, offset, nx, nn, 30, st);   // :433
            if (strn > 1.) strn = 1. / strn;                                  // :435
            if (!((freq == 0.0) || (freq > fmin && freq < fmax))) continue;   // :439
            double xmid, ymid;
            improve_extremum_sinc(ys, n, ylen, offset, nx, nn, 1200, xmid, ymid, st);   // :444
            xmid += (double)offset;                                           // :445
            if (ymid > 1.) ymid = 1. / ymid;                                  // :446
            const double cf = sample_rate / xmid, cs = ymid;                  // :447-448
            if (cs != cs) any_nan = true;
            // stable descending insertion == prefix of the reference's stable sort (:453)
            const int pos = __popcll(__ballot(lane < kept && ls >= cs));
            const double pf = from_prev_lane(lf), ps = from_prev_lane(ls);
            if (lane > pos) { lf = pf; ls = ps; }
            if (lane == pos) { lf = cf; ls = cs; }
            kept = (kept + 1 < kmax) ? kept + 1 : kmax;
            total_cand++;
        }
    }
    {   // maxima.push(Pitch::new(0, threshold)), :452
        const int pos = __popcll(__ballot(lane < kept && ls >= threshold));
        const double pf = from_prev_lane(lf), ps = from_prev_lane(ls);
        if (lane > pos) { lf = pf; ls = ps; }
        if (lane == pos) { lf = 0.0; ls = threshold; }
        kept = (kept + 1 < kmax) ? kept + 1 : kmax;
        total_cand++;
    }
    if (total_cand > 1 && (any_nan || threshold != threshold)) st |= 8;   // partial_cmp().unwrap() panics (Q10)
    int code = 0;
    if (st & 4) code = 4; else if (st & 8) code = 3;
    if (lane < kmax) {
        pitch_t o;
        const bool valid = (code == 0) && lane < kept;
        o.frequency = valid ? lf : 0.0;
        o.strength = valid ? ls : 0.0;
        out_cand[f * (long)kmax + lane] = o;
    }
    if (lane == 0) {
        if (out_count != nullptr) out_count[f] = (code == 0) ? total_cand : 0;
        if (status != nullptr) status[f] = code;
    }
}

// ------------------------------------------------------------------------------------------
// interpolate_sinc / improve_extremum at M query points of one curve (one wavefront per point)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sinc_points_kernel(const double *__restrict__ y, int ylen, long offset, long nx,
                                                         const double *__restrict__ xs, long m, long depth,
                                                         double *__restrict__ out, int32_t *__restrict__ status) {
    const long q = blockIdx.x;
    if (q >= m) return;
    int st = 0;
    const double v = sinc_interp(y, ylen, ylen, offset, nx, xs[q], depth, st);
    if (lane_id() == 0) {
        out[q] = (st & 4) ? 0.0 : v;
        if (status != nullptr) status[q] = (st & 4) ? 4 : 0;
    }
}

__global__ __launch_bounds__(64) void extremum_points_kernel(const double *__restrict__ y, int ylen, long offset, long nx,
                                                             const double *__restrict__ ix, long m, long depth,
                                                             double *__restrict__ out_xy, int32_t *__restrict__ status) {
    const long q = blockIdx.x;
    if (q >= m) return;
    int st = 0;
    double xmid, ymid;
    improve_extremum_sinc(y, ylen, ylen, offset, nx, ix[q], depth, xmid, ymid, st);
    if (lane_id() == 0) {
        out_xy[2 * q] = (st & 4) ? 0.0 : xmid;
        out_xy[2 * q + 1] = (st & 4) ? 0.0 : ymid;
        if (status != nullptr) status[q] = (st & 4) ? 4 : 0;
    }
}

size_t pitch_lds_bytes(int n) { return (size_t)(2 * n + autocorr_pad(n)) * sizeof(double); }

void launch_pitch(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                  const double *lag_window, double sample_rate, double threshold, double fmin, double fmax,
                  int kmax, pitch_t *out_cand, int32_t *out_count, int32_t *status) {
    hipLaunchKernelGGL(pitch_kernel, dim3((unsigned)F), dim3(64), pitch_lds_bytes(n), s,
                       x, F, n, stride, window, lag_window, sample_rate, threshold, fmin, fmax, kmax,
                       out_cand, out_count, status);
}

void launch_sinc_points(hipStream_t s, const double *y, int ylen, long offset, long nx, const double *xs, long m,
                        long depth, double *out, int32_t *status) {
    hipLaunchKernelGGL(sinc_points_kernel, dim3((unsigned)m), dim3(64), 0, s, y, ylen, offset, nx, xs, m, depth, out, status);
}

void launch_extremum_points(hipStream_t s, const double *y, int ylen, long offset, long nx, const double *ix, long m,
                            long depth, double *out_xy, int32_t *status) {
    hipLaunchKernelGGL(extremum_points_kernel, dim3((unsigned)m), dim3(64), 0, s, y, ylen, offset, nx, ix, m, depth, out_xy, status);
}

}  // namespace vbx
